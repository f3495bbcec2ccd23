#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# two ranks on one card (gloo): the N>1 flow of bench.py -- shared corpus files, index replica, strong scaling, chunked spool
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --single-device --config cfg4 --pairs 400000 --queries 6000 --chunk-queries 2000 --steps 2 --warmup 1 > gpurun_out/r2l_two_ranks.log 2> gpurun_out/r2l_two_ranks.err; echo "2-rank rc=$?"
tail -c 1500 gpurun_out/r2l_two_ranks.log; tail -3 gpurun_out/r2l_two_ranks.err
bash tools/kstats.sh r2l_kstats; echo "kstats rc=$?"
bash tools/pmc_passes.sh gpurun_out/r2l_pmc && cat gpurun_out/r2l_pmc/p*.sum.txt > gpurun_out/r2l_pmc_per_batch_kernels.txt; echo "pmc rc=$?"
