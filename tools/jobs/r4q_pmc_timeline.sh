#!/bin/bash
# round 4: counter passes over the lookups and the new kernels at the head (separate --pmc runs, no tracing), then the batch timeline (kernel trace only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PMC_KER='k_look1|k_look2|k_sort_lists|k_select_hits|k_select_rank|k_extract1|k_extract2' PMC_ONLY="1 3 4 7" bash tools/pmc_passes.sh gpurun_out/r4q_pmc > gpurun_out/r4q_pmc.log 2>&1
cat gpurun_out/r4q_pmc/p*.sum.txt > gpurun_out/r4q_pmc_lookup_extract_kernels.txt 2>/dev/null
grep -E "k_look|k_sort|k_select" gpurun_out/r4q_pmc_lookup_extract_kernels.txt | grep -E "TCC_EA0_RDREQ_sum|SQ_WAVES|SQ_INSTS_VALU |FETCH_SIZE|TCC_EA0_RDREQ_DRAM" | head -30
bash tools/jobs/gpu_timeline.sh r4q
sed -n 1,60p gpurun_out/r4q_timeline.txt
