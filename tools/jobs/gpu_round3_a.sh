#!/bin/bash
# GPU suite, then the driver's bench command and the toy line (round 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3a}
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${T}_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/${T}_pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_cfg3.json 2> gpurun_out/${T}_bench_cfg3.err; rc=$?
echo "bench rc=$rc"; head -c 1500 gpurun_out/${T}_bench_cfg3.json; echo
[ $rc -ne 0 ] && { tail -20 gpurun_out/${T}_bench_cfg3.err; exit $rc; }
timeout -k 10 300 python3 bench.py --config toy > gpurun_out/${T}_bench_toy.json 2> gpurun_out/${T}_bench_toy.err; rc=$?
echo "toy rc=$rc"; head -c 600 gpurun_out/${T}_bench_toy.json; echo
exit $rc
