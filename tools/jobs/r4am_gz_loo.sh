#!/bin/bash
# round 4: where the .gz formatter's time goes -- builds with one part left out (GZ_LOO, output meaningless), its two passes per batch on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 0 1 2 3 0; do
  if [ $n = 0 ]; then unset CGX_LIB; else export CGX_LIB=$GRAFT_REPO_ROOT/cgx_amd/libcgx_gzloo$n.so; fi
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fresh-steps 0 --gz-steps 3 --query-sets 1 > gpurun_out/r4am_l$n.log 2>gpurun_out/r4am_l$n.err || { tail -5 gpurun_out/r4am_l$n.err; continue; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4am_l$n.log") if l.startswith("{")][-1])
g=j["gz"]; print("GZ_LOO $n:", "count", g["fmt_count_ms_per_step"], "write", g["fmt_write_ms_per_step"], "format", g["format_ms_per_step"], "d2h", g["d2h_bytes_per_step"])
P
done
