#!/bin/bash
# host-side experiment: page-cache copy rate against the piece size (16 threads, in place)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; gcc -O2 -o /tmp/fa_bin tools/micro/file_assemble.c -lpthread || exit 1
D=/dev/shm/fa_$$; mkdir -p $D; OUT=gpurun_out/${1:-fa2}.txt; : > $OUT
for pm in 600 3000 12000 60000 400000; do /tmp/fa_bin $D 16 3000 3.3 8 1 2 $pm | tail -1 >> $OUT; done
rm -rf $D; cat $OUT
