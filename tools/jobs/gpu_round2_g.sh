#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/micro/gather_bw > gpurun_out/r2g_gather_bw.jsonl 2>&1; cat gpurun_out/r2g_gather_bw.jsonl
timeout -k 10 500 python3 bench.py --config cfg4 > gpurun_out/r2g_bench_cfg4.log 2> gpurun_out/r2g_bench_cfg4.err; echo "cfg4 rc=$?"
timeout -k 10 600 python3 bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r2g_bench_cfg5.log 2> gpurun_out/r2g_bench_cfg5.err; echo "cfg5 rc=$?"
python3 - <<P
import json
for f in ("gpurun_out/r2g_bench_cfg4.log","gpurun_out/r2g_bench_cfg5.log"):
    for line in open(f):
        if line.startswith("{"):
            d=json.loads(line); print(f, d["value"], d["ms_per_step"], d["config"]["outdir_mode"], d["hbm_in_use_gb"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["index"])
P
tail -3 gpurun_out/r2g_bench_cfg5.err
