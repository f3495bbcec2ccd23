#!/bin/bash
# round 4: dynamic Huffman codes per batch: gz tests, then the gz leg of cfg3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gzip or staged or bit_exact or async or every_stage" > gpurun_out/r4o_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/r4o_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 10 > gpurun_out/r4o_cfg3.log 2>gpurun_out/r4o_cfg3.err || { tail -20 gpurun_out/r4o_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4o_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print({k: v for k, v in j["gz"].items() if k not in ("stages_ms_per_step", "note")})
P
