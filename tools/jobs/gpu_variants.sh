#!/bin/bash
# A/B runs of bench.py --no-write with cgx_set_option overrides; prints the lookup stage times of each variant.
# usage: tools/jobs/gpu_variants.sh <tag> "<opts of variant 1>" "<opts of variant 2>" ...   (opts = space separated name=value, "-" = defaults)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
i=0
for v in "$@"; do
  i=$((i+1)); args=""
  if [ "$v" != "-" ]; then for o in $v; do args="$args --option $o"; done; fi
  timeout -k 10 300 python3 bench.py --no-write --no-cpu-baseline --steps 3 --warmup 1 $args > gpurun_out/${TAG}_v$i.log 2> gpurun_out/${TAG}_v$i.err || echo "variant $i failed"
  python3 - <<P
import json
for line in open("gpurun_out/${TAG}_v$i.log"):
    if line.startswith("{"):
        d = json.loads(line); s = d["stages_ms_per_step"]
        print("variant $i [$v]: look1 %.2f look2 %.2f gappy %.1f extract %.1f lexicon %.1f total_gpu %.1f ms/step %.1f" % (s["look1_kernel"], s["look2_kernel"], s["gappy"], s["extract"], s["lexicon"], s["host_total"], d["ms_per_step"]))
P
done
