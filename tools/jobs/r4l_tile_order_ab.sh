#!/bin/bash
# round 4: k_look1's tiles in corpus-region order (tile_order 1) against group order (0), same box, no files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for o in 0 1 0 1; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write --option tile_order=$o > gpurun_out/r4l_to$o.log 2>gpurun_out/r4l_to$o.err || { tail -20 gpurun_out/r4l_to$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4l_to$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("tile_order $o:", j["ms_per_step"], "gappy", s["gappy"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "extract", s["extract"], "select", s.get("select_hits"))
P
done
