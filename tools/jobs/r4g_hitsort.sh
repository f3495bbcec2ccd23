#!/bin/bash
# round 4: hit lists ordered by pattern + position bucket, order statistics selected (k_select_hits): parity first, then cfg3 without files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "every_stage or definitions or bit_exact or sizing or id_level" > gpurun_out/r4g_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/r4g_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
CGX_TRACE_SELECT=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4g_cfg3_nowrite.log 2>gpurun_out/r4g_cfg3_nowrite.err || { tail -20 gpurun_out/r4g_cfg3_nowrite.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4g_cfg3_nowrite.log") if l.startswith("{")][-1])
print("cfg3 no-write", j["value"], j["ms_per_step"], j.get("report_error"))
print({k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_")})
P
