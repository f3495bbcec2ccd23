#!/bin/bash
# round 4 (re-entry): the whole GPU suite at the head, then cfg3 with the gz leg (no CPU baseline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r4d_pytest.log 2>&1; rc=$?
tail -25 gpurun_out/r4d_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fresh-steps 0 > gpurun_out/r4d_cfg3.log 2>gpurun_out/r4d_cfg3.err || { tail -20 gpurun_out/r4d_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4d_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print(json.dumps(j["gz"], indent=1))
print(j["per_rank"]); print(j["hbm_breakdown"]); print({k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_t_")})
print("reruns", j["append_pass_reruns_per_step"], "dominant", j["roofline_dominant"]["kernel"])
P
