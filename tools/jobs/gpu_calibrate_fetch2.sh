#!/bin/bash
# second calibration pass: does TCC_EA0_RDREQ_DRAM_32B count 32-byte units of the bytes actually read (known-byte gather kernels)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-cal2}; mkdir -p $OUT
[ -x bin/gather_bw_micro ] || { mkdir -p bin && hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.cpp -o bin/gather_bw_micro; } || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum TCC_READ_sum --kernel-include-regex "k_gather" --output-format csv -d $OUT/p -- $GRAFT_REPO_ROOT/bin/gather_bw_micro > $OUT/p.log 2>&1 || echo "pass failed"
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" > $OUT/p.txt <<Q
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Kernel_Name"][:34], "grid", r.get("Grid_Size"), r["Counter_Name"], r["Counter_Value"])
Q
rm -rf $OUT/p; grep -v "grid 16777216\|once" $OUT/p.txt | awk 'NR%1==0' | head -60
