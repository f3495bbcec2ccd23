#!/bin/bash
# round 4, head after the MaxLex task list and the lookup trims: the whole GPU suite, then the driver's bench command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r4bi_pytest.log 2>&1; rc=$?
tail -16 gpurun_out/r4bi_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4bi_bench_line_cfg3.json 2>gpurun_out/r4bi_bench.err; rc=$?
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4bi_bench_line_cfg3.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "gz", j["value_gz"], "fresh", j["value_fresh_files"], "chain", j["value_gpu_chain"], j.get("report_error"))
print("gz", {k: v for k, v in j["gz"].items() if k not in ("stages_ms_per_step", "note")})
print("stages", {k:v for k,v in j["stages_ms_per_step"].items() if not k.startswith("host_t")})
P
exit $rc
