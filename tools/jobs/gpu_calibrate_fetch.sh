#!/bin/bash
# Calibration of FETCH_SIZE / TCC_EA0_RDREQ for the lookups' access pattern (every lane reads 128 bytes with eight 16-byte
# loads at a random offset): tools/micro/gather_bw reads a KNOWN number of bytes that way; the guide says a wide coalesced
# stream is tallied at half its bytes on gfx950 and that other patterns must be calibrated.  One rocprofv3 --pmc pass per group.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-cal}; mkdir -p $OUT
[ -x bin/gather_bw_micro ] || { mkdir -p bin && hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.cpp -o bin/gather_bw_micro; } || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_HIT_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "k_gather" --output-format csv -d $OUT/p$i -- $GRAFT_REPO_ROOT/bin/gather_bw_micro > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $OUT/p$i.txt <<Q
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(r["Kernel_Name"][:60], "grid", r.get("Grid_Size"), r["Counter_Name"], r["Counter_Value"])
Q
  rm -rf $OUT/p$i
done
cat $OUT/p1.txt $OUT/p2.txt
