#!/bin/bash
# round 4: two bench RANKS on one card (--single-device, gloo; each with its own index), every step written as .gz, against one rank alone
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 0 --option gz_level=1 > gpurun_out/r4bc_one.log 2>gpurun_out/r4bc_one.err || { tail -5 gpurun_out/r4bc_one.err; exit 1; }
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --single-device --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 0 --option gz_level=1 > gpurun_out/r4bc_two.log 2>gpurun_out/r4bc_two.err || { tail -8 gpurun_out/r4bc_two.err; exit 1; }
python - <<P
import json
for f in ("one","two"):
    j=json.loads([l for l in open("gpurun_out/r4bc_%s.log" % f) if l.startswith("{")][-1])
    print(f, "rank(s) on one card, .gz files:", j["value"], "queries/s,", j["ms_per_step"], "ms per step; chain", j["per_rank"]["gpu_chain_ms_per_step"], "file", j["per_rank"]["file_phase_ms_per_step"], "dma", j["per_rank"]["dma_wait_ms_per_step"], "hbm", j.get("hbm_in_use_gb"))
P
