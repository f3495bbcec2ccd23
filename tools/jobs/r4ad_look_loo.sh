#!/bin/bash
# round 4: where k_look1's time goes -- builds with one part left out (CGX_LOO, results meaningless), the kernel's time per batch on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 0 1 2 3 4 0; do
  if [ $n = 0 ]; then unset CGX_LIB; else export CGX_LIB=$GRAFT_REPO_ROOT/cgx_amd/libcgx_loo$n.so; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-write --query-sets 1 > gpurun_out/r4ad_l$n.log 2>gpurun_out/r4ad_l$n.err || { tail -5 gpurun_out/r4ad_l$n.err; continue; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ad_l$n.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("CGX_LOO $n:", "look1", s["look1_kernel"], "look2", s["look2_kernel"], "gappy", s["gappy"], "hits", j.get("counts",{}).get("hits1"), j.get("counts",{}).get("hits2"))
P
done
