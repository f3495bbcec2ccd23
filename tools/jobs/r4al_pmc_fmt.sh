#!/bin/bash
# round 4: vector / scalar instruction counts of the .gz formatter passes after the digit table (separate --pmc runs, no tracing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PMC_WRITE=1 PMC_KER='k_fmt_lines' PMC_ONLY="1 2" bash tools/pmc_passes.sh gpurun_out/r4al_pmc --gz-steps 0 --query-sets 1 --option gz_level=1 > gpurun_out/r4al_pmc.log 2>&1
cat gpurun_out/r4al_pmc/p*.sum.txt > gpurun_out/r4al_pmc_fmt_gz.txt 2>/dev/null
grep -E "SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_LDS|GRBM_GUI|SQ_WAIT_ANY|SQ_WAVE_CYCLES" gpurun_out/r4al_pmc_fmt_gz.txt | sed 's/  */ /g'
