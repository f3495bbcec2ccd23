#!/bin/bash
# the whole cfg5 step (10^6 queries, files written) with two builds of the library, one after the other.  usage: tools/jobs/gpu_cfg5_ab.sh <tag> <lib> <lib>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  CGX_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 bench.py --config cfg5 --no-cpu-baseline --fresh-steps 0 > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { echo "$name failed"; exit 1; }
  python3 - <<P
import json
for line in open("gpurun_out/${TAG}_$name.json"):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]; k = [d["roofline_dominant"]] + d["roofline_other_kernels"]
        print("$name: value %.0f gpu chain %.0f; gappy %.0f extract %.0f lexicon %.0f format %.0f; per batch:" % (d["value"], d["value_gpu_chain"], s["gappy"], s["extract"], s["lexicon"], s["format"]), [(e["kernel"][:7], e["ms_per_launch"]) for e in k])
P
done
