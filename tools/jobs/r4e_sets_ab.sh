#!/bin/bash
# round 4: is the file phase of the rotating bench slower because the files change size (query sets 3 vs 1), or is it the box?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for sets in 1 3; do
  CGX_TRACE=1 timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 4 --query-sets $sets > gpurun_out/r4e_sets$sets.log 2>gpurun_out/r4e_sets$sets.err || { tail -20 gpurun_out/r4e_sets$sets.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4e_sets$sets.log") if l.startswith("{")][-1])
print("sets $sets: value", j["value"], "ms", j["ms_per_step"], "per_rank", j["per_rank"]["dma_wait_ms_per_step"], j["per_rank"]["file_phase_ms_per_step"])
g=j["gz"]; print("   gz", g["value"], g["ms_per_step"], "chain", g["gpu_chain_ms_per_step"], "dma", g["dma_wait_ms_per_step"], "file", g["file_phase_ms_per_step"], "d2h", g["d2h_bytes_per_step"], "files", g["gz_file_bytes_per_step"])
P
  grep "cgx writer" gpurun_out/r4e_sets$sets.err | tail -4
done
