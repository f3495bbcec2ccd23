#!/bin/bash
# first GPU call of round 2: parity suite, then the driver's exact bench command, then one diagnostic step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a_bench.log 2> gpurun_out/r2a_bench.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r2a_bench.log
tail -5 gpurun_out/r2a_bench.err
