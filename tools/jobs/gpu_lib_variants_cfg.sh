#!/bin/bash
# as gpu_lib_variants.sh, with the bench arguments given: tools/jobs/gpu_lib_variants_cfg.sh <tag> "<bench args>" <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; ARGS=$2; shift; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  CGX_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python3 bench.py --no-write --no-cpu-baseline $ARGS > gpurun_out/${TAG}_$name.log 2> gpurun_out/${TAG}_$name.err || echo "$name failed"
  python3 - <<P
import json
for line in open("gpurun_out/${TAG}_$name.log"):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]; k = [d["roofline_dominant"]] + d["roofline_other_kernels"]
        print("$name: look1 %.2f look2 %.2f gappy %.1f extract %.1f lexicon %.1f per step; per batch:" % (s["look1_kernel"], s["look2_kernel"], s["gappy"], s["extract"], s["lexicon"]), [(e["kernel"][:7], e["occurrences_per_launch"], e["ms_per_launch"]) for e in k])
P
done
