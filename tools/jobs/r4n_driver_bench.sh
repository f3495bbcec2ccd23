#!/bin/bash
# round 4: the driver's bench command at the head (CPU baseline, fresh-files and gz legs included); the line goes to gpurun_out
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4n_bench_line_cfg3.json 2>gpurun_out/r4n_bench.err; rc=$?
tail -3 gpurun_out/r4n_bench.err
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4n_bench_line_cfg3.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "gz", j["value_gz"], "fresh", j["value_fresh_files"], "chain", j["value_gpu_chain"], j.get("report_error"))
print("roofline", j["roofline"]["frac"], "dominant", j["roofline_dominant"]["kernel"], j["roofline_dominant"]["frac"])
print("cpu", {k: (v if not isinstance(v, dict) else "...") for k, v in j["cpu_baseline"].items()})
print("gz", {k: v for k, v in j["gz"].items() if k not in ("stages_ms_per_step", "note")})
P
exit $rc
