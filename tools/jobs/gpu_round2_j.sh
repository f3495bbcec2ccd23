#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -C oracle strmatch_oracle liboracle.so > /dev/null 2>&1
timeout -k 10 700 python3 tools/stress_parity.py --fuzz 60 --seed 41 > gpurun_out/r2j_fuzz41.log 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r2j_fuzz41.log | cut -c1-250; grep -c OK gpurun_out/r2j_fuzz41.log; grep -c MISMATCH gpurun_out/r2j_fuzz41.log
timeout -k 10 400 python3 tools/stress_parity.py --fuzz 6 --seed 42 --big > gpurun_out/r2j_fuzzbig42.log 2>&1; echo "fuzzbig rc=$?"; grep -c OK gpurun_out/r2j_fuzzbig42.log
