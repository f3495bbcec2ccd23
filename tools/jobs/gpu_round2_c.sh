#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CGX_DIAG_GROUPS=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-write --no-cpu-baseline > gpurun_out/r2c_diag.log 2> gpurun_out/r2c_diag.err; echo "diag rc=$?"
grep -A16 "look1 groups" gpurun_out/r2c_diag.err | head -40
bash tools/kstats.sh r2c_kstats; echo "kstats rc=$?"
bash tools/pmc_sa_lookup.sh gpurun_out/r2c_pmc_sa; echo "pmc rc=$?"
cat gpurun_out/r2c_pmc_sa/pmc_k_sa_lookup.txt
