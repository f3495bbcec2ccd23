#!/bin/bash
# round 4, head: counter passes over the lookups, extraction, MaxLex and run_sort kernels (separate --pmc runs, no tracing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PMC_KER='k_look1|k_look2|k_extract|k_lex_finish|k_runsort_block|k_sort_lists|k_select' PMC_ONLY="1 2 3 4" bash tools/pmc_passes.sh gpurun_out/r4ai_pmc > gpurun_out/r4ai_pmc.log 2>&1
cat gpurun_out/r4ai_pmc/p*.sum.txt > gpurun_out/r4ai_pmc_head_kernels.txt 2>/dev/null
grep -E "k_look1|k_look2" gpurun_out/r4ai_pmc_head_kernels.txt | grep -E "SQ_INSTS_VALU |SQ_INSTS_SALU|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|GRBM_GUI" | awk '{printf "%-22s %-26s n=%s total=%s\n", $1" "$2, $3, $5, $7}'
