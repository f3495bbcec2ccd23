#!/bin/bash
# round 4: gz tests, then cfg3 with the gz leg (no CPU baseline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gzip or staged or bit_exact" > gpurun_out/r4c_pytest.log 2>&1 || { tail -30 gpurun_out/r4c_pytest.log; exit 1; }
tail -2 gpurun_out/r4c_pytest.log
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fresh-steps 0 > gpurun_out/r4c_cfg3.log 2>gpurun_out/r4c_cfg3.err || { tail -20 gpurun_out/r4c_cfg3.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4c_cfg3.log") if l.startswith("{")][-1])
print("cfg3", j["value"], j["ms_per_step"], "gz", j["value_gz"], j.get("report_error"))
print(json.dumps(j["gz"], indent=1))
P
