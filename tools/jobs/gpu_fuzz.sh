#!/bin/bash
# randomized oracle-vs-GPU parity sweep.  usage: tools/jobs/gpu_fuzz.sh <tag> <seed> [cases] [big-seed] [big-cases]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-fuzz}; SEED=${2:-51}; N=${3:-60}
make -C oracle strmatch_oracle liboracle.so > /dev/null 2>&1
timeout -k 10 800 python3 tools/stress_parity.py --fuzz $N --seed $SEED > gpurun_out/${TAG}_fuzz${SEED}.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -3 gpurun_out/${TAG}_fuzz${SEED}.log | cut -c1-250; grep -c OK gpurun_out/${TAG}_fuzz${SEED}.log; grep -c MISMATCH gpurun_out/${TAG}_fuzz${SEED}.log
[ $rc -ne 0 ] && exit $rc
if [ -n "$4" ]; then timeout -k 10 500 python3 tools/stress_parity.py --fuzz ${5:-6} --seed $4 --big > gpurun_out/${TAG}_fuzzbig$4.log 2>&1; echo "fuzzbig rc=$?"; grep -c OK gpurun_out/${TAG}_fuzzbig$4.log; grep -c MISMATCH gpurun_out/${TAG}_fuzzbig$4.log; fi
exit 0
