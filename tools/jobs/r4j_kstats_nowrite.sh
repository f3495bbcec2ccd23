#!/bin/bash
# round 4: per-kernel times of cfg3 without files (the GPU chain alone)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-write > $GRAFT_REPO_ROOT/gpurun_out/r4j_kstats.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4j_kstats_nowrite.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
python3 - <<P
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4j_kstats_nowrite.csv")))
for r in rows[:48]:
    print(r['Name'].split('(')[0].replace('void ','')[:70].ljust(70), r['Calls'], round(float(r["TotalDurationNs"])/1e6/8,2), round(float(r['AverageNs'])/1e6,3))
P
