#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench; copies the per-kernel csv to gpurun_out/<name>.csv
NAME=${1:-kstats}; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$NAME.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/$NAME.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
python3 - <<P
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/$NAME.csv")))
for r in rows[:22]:
    print(r['Name'].split('(')[0].replace('void ','')[:60].ljust(60), r['Calls'], round(float(r["TotalDurationNs"])/1e6/8,1), round(float(r['AverageNs'])/1e6,2))
P
