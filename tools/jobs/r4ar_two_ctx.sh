#!/bin/bash
# round 4: the bench's .gz leg with one context and with two contexts over one index
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 8 --two-context-steps 12 > gpurun_out/r4ar_bench.log 2>gpurun_out/r4ar_bench.err || { tail -20 gpurun_out/r4ar_bench.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ar_bench.log") if l.startswith("{")][-1])
g=j["gz"]; print("gz one context:", g["value"], g["ms_per_step"], "chain", g["gpu_chain_ms_per_step"], "dma", g["dma_wait_ms_per_step"], "file", g["file_phase_ms_per_step"])
print("gz two contexts:", json.dumps(j["gz_two_contexts"]))
P
