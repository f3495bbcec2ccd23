#!/bin/bash
# A/B: writer thread count under the 16-CPU quota (the main thread that feeds the GPU shares the quota with the writers)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-thr}; shift
for t in "$@"; do
  CGX_THREADS=$t timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_t$t.log 2> gpurun_out/${TAG}_t$t.err; echo "threads $t rc=$?"
  python3 - <<Q
import json
for line in open("gpurun_out/${TAG}_t$t.log"):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]
        print("  threads $t:", d["value"], d["ms_per_step"], "gpu", round(s["gappy"] + s["extract"] + s["lexicon"] + s["format"], 1), "d2h", s["host_write_wait_d2h"], "file", s["host_write_file"], "host_total", s["host_total"])
Q
done
grep -c . /proc/pressure/cpu 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
