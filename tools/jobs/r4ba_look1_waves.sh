#!/bin/bash
# round 4: k_look2 at five waves per SIMD (80 registers, 64 bytes of scratch per lane) against five (93 registers), same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 0 1 0 1; do
  if [ $n = 0 ]; then unset CGX_LIB; else export CGX_LIB=$GRAFT_REPO_ROOT/cgx_amd/libcgx_l1w5.so; fi
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4ba_p$n.log 2>gpurun_out/r4ba_p$n.err || { tail -5 gpurun_out/r4ba_p$n.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ba_p$n.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("five waves $n:", j["ms_per_step"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "gappy", s["gappy"])
P
done
