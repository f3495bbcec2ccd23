#!/bin/bash
# round 4: two contexts over one index: do their compute streams share a hardware queue?  The .gz legs with the compute streams at normal priority,
# and with more hardware queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "CGX_COMPUTE_PRIORITY=normal" "CGX_COMPUTE_PRIORITY=normal GPU_MAX_HW_QUEUES=8"; do
  env $v timeout -k 10 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fresh-steps 0 --gz-steps 8 --two-context-steps 12 > gpurun_out/r4as_bench.log 2>gpurun_out/r4as_bench.err || { tail -20 gpurun_out/r4as_bench.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4as_bench.log") if l.startswith("{")][-1])
g=j["gz"]; t=j["gz_two_contexts"]
print("$v | one context:", g["value"], g["ms_per_step"], "chain", g["gpu_chain_ms_per_step"], "| two contexts:", t.get("value"), t.get("ms_per_step"), t.get("gpu_chain_ms_per_batch_while_sharing_the_card"), t.get("note") if not t.get("value") else "")
P
done
