#!/bin/bash
# host-side experiment: fresh files vs first rewrite vs later rewrites (14 threads, 5 KB pieces), pwritev only
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; gcc -O2 -o /tmp/fa_bin tools/micro/file_assemble.c -lpthread || exit 1
D=/dev/shm/fa_$$; mkdir -p $D; /tmp/fa_bin $D 14 6000 3.3 8 1 0 10000 > gpurun_out/${1:-fa3}.txt; rm -rf $D; cat gpurun_out/${1:-fa3}.txt
