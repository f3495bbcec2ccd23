#!/bin/bash
# round 4: the new tests (run_sort with long runs, RCCL on a one-rank communicator, bench under torch.distributed nccl), then the gz leg with its own stage breakdown
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "run_sort or rccl or one_rank or query_limit or query_token_limit or every_stage" > gpurun_out/r4f_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r4f_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 5 > gpurun_out/r4f_cfg3.log 2>gpurun_out/r4f_cfg3.err || { tail -20 gpurun_out/r4f_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4f_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print(j["per_rank"]); print({k:v for k,v in j["stages_ms_per_step"].items()})
print(json.dumps(j["gz"], indent=1))
P
