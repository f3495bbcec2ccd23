#!/bin/bash
# the other presets' bench lines at the head: toy, cfg4, cfg5 (the driver's cfg3 command is tools/jobs/gpu_tests_bench.sh).  usage: tools/jobs/gpu_round3_h.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3h}
timeout -k 10 200 python3 bench.py --config toy --no-cpu-baseline > gpurun_out/${T}_bench_toy.json 2> gpurun_out/${T}_bench_toy.err; echo "toy rc=$?"; head -c 300 gpurun_out/${T}_bench_toy.json; echo
timeout -k 10 300 python3 bench.py --config cfg4 --no-cpu-baseline > gpurun_out/${T}_bench_cfg4.json 2> gpurun_out/${T}_bench_cfg4.err || exit 1; echo "cfg4 rc=0"; head -c 300 gpurun_out/${T}_bench_cfg4.json; echo
timeout -k 10 400 python3 bench.py --config cfg5 --no-cpu-baseline --fresh-steps 0 > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err; echo "cfg5 rc=$?"; head -c 300 gpurun_out/${T}_bench_cfg5.json; echo
