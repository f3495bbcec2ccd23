#!/bin/bash
# A/B: do the D2H text copies run on the SDMA engines or as blit kernels (__amd_rocclr_copyBuffer), with and without the profiler?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-sdma}
show() { python3 - "$1" <<Q
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]; r = d["roofline"]
        print("   ", d["value"], d["ms_per_step"], "k_sa_lookup ms (events, mean of timed steps)", r["kernel_ms_mean_timed_steps"], "gpu", round(s["gappy"] + s["extract"] + s["lexicon"] + s["format"], 1), "d2h", s["host_write_wait_d2h"], "file", s["host_write_file"])
Q
}
echo "plain run"; timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_plain.log 2>&1; show gpurun_out/${TAG}_plain.log
echo "HSA_ENABLE_SDMA=0 (copies as blit kernels)"; HSA_ENABLE_SDMA=0 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_nosdma.log 2>&1; show gpurun_out/${TAG}_nosdma.log
cd /tmp && export TMPDIR=/tmp
echo "under rocprofv3 --kernel-trace --stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sd_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1; show $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/sd_tmp -name "*kernel_stats.csv" | head -1); grep -i "copyBuffer\|k_sa_lookup<false>" "$f" | cut -c1-160; rm -rf $GRAFT_REPO_ROOT/gpurun_out/sd_tmp
