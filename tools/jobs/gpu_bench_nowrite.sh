#!/bin/bash
# GPU chain only (rules counted on the device, nothing leaves the card): stage times of a short run.  usage: tools/jobs/gpu_bench_nowrite.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-nw}
timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-write --no-cpu-baseline > gpurun_out/${TAG}_nowrite.log 2> gpurun_out/${TAG}_nowrite.err; echo "rc=$?"
python3 - <<Q
import json
for line in open("gpurun_out/${TAG}_nowrite.log"):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], {k: v for k, v in d["stages_ms_per_step"].items() if v and not k.startswith("host_")})
Q
