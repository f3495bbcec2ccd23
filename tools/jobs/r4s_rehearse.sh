#!/bin/bash
# round 4: what one host writes when 1/2/4/8 ranks write at once -- one recorded cfg3 batch of 2 500 queries, as .gz pieces and as plain text.
# chain-ms: the GPU chain of such a batch on one card (a quarter of the 10 000-query batch: 240 ms .gz / 217 ms plain)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python tools/rehearse_writers.py --queries 2500 --ranks 1,2,4,8 --gz --chain-ms 60 --dir /dev/shm/cgx_rehearse_gz > gpurun_out/r4s_writer_rehearsal_gz.txt 2>gpurun_out/r4s_gz.err || { tail -5 gpurun_out/r4s_gz.err; exit 1; }
grep -E "^R=|host:" gpurun_out/r4s_writer_rehearsal_gz.txt
timeout -k 10 500 python tools/rehearse_writers.py --queries 2500 --ranks 1,8 --chain-ms 54 --dir /dev/shm/cgx_rehearse_plain > gpurun_out/r4s_writer_rehearsal_plain.txt 2>gpurun_out/r4s_plain.err || { tail -5 gpurun_out/r4s_plain.err; exit 1; }
grep -E "^R=|host:" gpurun_out/r4s_writer_rehearsal_plain.txt
