#!/bin/bash
# host-side experiment on the GPU box: page-cache copy rate of the file phase against thread count and placement
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; gcc -O2 -o /tmp/fa tools/micro/file_assemble.c -lpthread || exit 1
D=/dev/shm/fa_$$; mkdir -p $D; OUT=gpurun_out/${1:-fa}.txt; : > $OUT
for cfg in "16 0" "16 1" "16 2" "16 4" "16 8" "24 0" "32 0" "32 2" "32 4" "48 0" "64 0" "64 2"; do
  set -- $cfg
  /tmp/fa $D $1 3000 3.3 8 1 $2 | tail -1 >> $OUT
done
rm -rf $D; cat $OUT
