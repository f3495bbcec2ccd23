#!/bin/bash
# the tracked profiles of a round: kernel-trace stats and the PMC passes at HEAD.  usage: tools/jobs/gpu_profiles.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-prof}
bash tools/kstats.sh ${TAG}_kstats; echo "kstats rc=$?"
bash tools/pmc_passes.sh gpurun_out/${TAG}_pmc && cat gpurun_out/${TAG}_pmc/p*.sum.txt > gpurun_out/${TAG}_pmc_per_batch_kernels.txt; echo "pmc rc=$?"
bash tools/pmc_sa_lookup.sh gpurun_out/${TAG}_pmc_sa; echo "pmc sa rc=$?"
