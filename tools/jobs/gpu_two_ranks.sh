#!/bin/bash
# two ranks on one card (gloo): the N>1 flow of bench.py -- shared corpus files, index replica, strong and weak scaling, chunked spool
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-two}
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --single-device --config cfg4 --pairs 400000 --queries 6000 --chunk-queries 2000 --steps 2 --warmup 1 > gpurun_out/${TAG}_two_ranks_strong.log 2> gpurun_out/${TAG}_two_ranks_strong.err; rc=$?; echo "2-rank strong rc=$rc"
tail -c 600 gpurun_out/${TAG}_two_ranks_strong.log; [ $rc -ne 0 ] && tail -5 gpurun_out/${TAG}_two_ranks_strong.err && exit $rc
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --backend gloo --single-device --pairs 400000 --queries 3000 --steps 3 --warmup 1 > gpurun_out/${TAG}_two_ranks_weak.log 2> gpurun_out/${TAG}_two_ranks_weak.err; rc=$?; echo "2-rank weak rc=$rc"
tail -c 600 gpurun_out/${TAG}_two_ranks_weak.log; [ $rc -ne 0 ] && tail -5 gpurun_out/${TAG}_two_ranks_weak.err
exit $rc
