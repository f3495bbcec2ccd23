#!/bin/bash
# round 4: deflate pieces v2 (groups as blocks, bit-granular lines, one gzip member per file) + hit selection from LDS: parity, then cfg3 no-write and the gz leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "every_stage or definitions or bit_exact or gzip or staged or async or sizing" > gpurun_out/r4h_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/r4h_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4h_cfg3_nowrite.log 2>gpurun_out/r4h_cfg3_nowrite.err || { tail -20 gpurun_out/r4h_cfg3_nowrite.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4h_cfg3_nowrite.log") if l.startswith("{")][-1])
print("cfg3 no-write", j["value"], j["ms_per_step"], j.get("report_error"))
print({k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_")})
P
timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 8 > gpurun_out/r4h_cfg3.log 2>gpurun_out/r4h_cfg3.err || { tail -20 gpurun_out/r4h_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4h_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print(json.dumps(j["gz"], indent=1))
P
