#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r2b_pytest.log
[ $rc -ne 0 ] && exit 1
CGX_TRACE=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r2b_trace.log 2> gpurun_out/r2b_trace.err; echo "trace rc=$?"
grep "cgx writer" gpurun_out/r2b_trace.err | tail -6
tail -c 3000 gpurun_out/r2b_trace.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2b_bench.log 2> gpurun_out/r2b_bench.err; echo "bench rc=$?"
tail -c 7000 gpurun_out/r2b_bench.log
