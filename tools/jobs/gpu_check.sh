#!/bin/bash
# GPU parity tests, then (only if green) the default bench; prints the bench line's headline numbers.
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
grep -q " passed" gpurun_out/t.log && ! grep -q "failed" gpurun_out/t.log || exit 1
timeout -k 10 400 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-rewrite-run "$@" > gpurun_out/b1.log 2>&1
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/b1.log") if l.startswith("{")][-1]); print(j["value"], j["ms_per_step"]); print({k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_t_")})
P
