#!/bin/bash
# round 4: the .gz writing pass at six waves per SIMD (80 registers, eight spilled) instead of five (94): the format stages of the .gz leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 4 > gpurun_out/r4z_bench.log 2>gpurun_out/r4z_bench.err || { tail -20 gpurun_out/r4z_bench.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4z_bench.log") if l.startswith("{")][-1])
g=j["gz"]; print("gz:", g["value"], g["ms_per_step"], "chain", g["gpu_chain_ms_per_step"], "format", g["format_ms_per_step"], "count", g["fmt_count_ms_per_step"], "write", g["fmt_write_ms_per_step"], "d2h", g["d2h_bytes_per_step"])
P
