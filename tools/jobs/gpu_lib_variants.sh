#!/bin/bash
# A/B of several builds of the library on one box: bench.py --no-write with CGX_LIB = each build in turn; prints the lookup
# kernels' and the stages' times.  usage: tools/jobs/gpu_lib_variants.sh <tag> <lib> [<lib> ...]   (paths relative to the repo root)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  CGX_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 bench.py --no-write --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/${TAG}_$name.log 2> gpurun_out/${TAG}_$name.err || echo "$name failed"
  python3 - <<P
import json
for line in open("gpurun_out/${TAG}_$name.log"):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]
        print("$name: look1 %.2f look2 %.2f gappy %.1f extract %.1f lexicon %.1f gpu %.1f ms/step %.1f rules/s %.4g" % (s["look1_kernel"], s["look2_kernel"], s["gappy"], s["extract"], s["lexicon"], s["host_total"], d["ms_per_step"], d["rules_per_s"]))
P
done
