#!/bin/bash
# host-side experiment: does a huge-page source text speed the page-cache copy up? (14 threads, 5 KB pieces, in place)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; gcc -O2 -o /tmp/fa_bin tools/micro/file_assemble.c -lpthread || exit 1
D=/dev/shm/fa_$$; mkdir -p $D; OUT=gpurun_out/${1:-fa4}.txt; : > $OUT
for thp in 0 1 0 1; do /tmp/fa_bin $D 14 6000 3.3 16 1 0 10000 $thp | tail -1 | sed "s/^/thp=$thp  /" >> $OUT; grep -i "AnonHugePages" /proc/meminfo >> $OUT; done
rm -rf $D; cat $OUT
