#!/bin/bash
# round 4: small bench to shake out the new legs, then cfg3 with the gz leg (no CPU baseline), line to gpurun_out
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --pairs 200000 --queries 500 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4b_small.log 2>gpurun_out/r4b_small.err || { tail -20 gpurun_out/r4b_small.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4b_small.log") if l.startswith("{")][-1])
print("small", j["value"], j["value_gz"], j.get("report_error"), j["gz"])
P
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4b_cfg3.log 2>gpurun_out/r4b_cfg3.err || { tail -20 gpurun_out/r4b_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4b_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print(json.dumps(j["gz"], indent=1))
print(j["per_rank"]); print(j["hbm_breakdown"]); print({k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_t_")})
print("reruns", j["append_pass_reruns_per_step"], "dominant", j["roofline_dominant"]["kernel"])
P
