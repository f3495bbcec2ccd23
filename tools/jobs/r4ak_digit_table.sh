#!/bin/bash
# round 4: word ids and decimals through the three-digit table: parity, then the format stages of the plain and the .gz leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4ak_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r4ak_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 4 > gpurun_out/r4ak_bench.log 2>gpurun_out/r4ak_bench.err || { tail -20 gpurun_out/r4ak_bench.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ak_bench.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("plain:", j["value"], j["ms_per_step"], "chain", j["per_rank"]["gpu_chain_ms_per_step"], {k:s[k] for k in ("gappy","extract","lexicon","format","fmt_count","fmt_write") if k in s})
g=j["gz"]; print("gz:", g["value"], g["ms_per_step"], "chain", g["gpu_chain_ms_per_step"], "format", g["format_ms_per_step"], "count", g["fmt_count_ms_per_step"], "write", g["fmt_write_ms_per_step"], "d2h", g["d2h_bytes_per_step"])
P
