#!/bin/bash
# the bench lines of the round: the driver's command, the toy line, cfg4, cfg5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3r}
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_cfg3.json 2> gpurun_out/${T}_bench_cfg3.err; echo "cfg3 rc=$?"; head -c 700 gpurun_out/${T}_bench_cfg3.json; echo
timeout -k 10 300 python3 bench.py --config toy > gpurun_out/${T}_bench_toy.json 2> gpurun_out/${T}_bench_toy.err; echo "toy rc=$?"; head -c 400 gpurun_out/${T}_bench_toy.json; echo
timeout -k 10 600 python3 bench.py --config cfg4 --no-cpu-baseline > gpurun_out/${T}_bench_cfg4.json 2> gpurun_out/${T}_bench_cfg4.err; echo "cfg4 rc=$?"; head -c 400 gpurun_out/${T}_bench_cfg4.json; echo
timeout -k 10 900 python3 bench.py --config cfg5 --no-cpu-baseline --fresh-steps 0 > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err; echo "cfg5 rc=$?"; head -c 400 gpurun_out/${T}_bench_cfg5.json; echo
