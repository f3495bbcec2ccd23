#!/bin/bash
# round 4: per-kernel times with every step written as .gz (dynamic codes, symbols made once)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --fresh-steps 0 --gz-steps 0 --option gz_level=1 > $GRAFT_REPO_ROOT/gpurun_out/r4r_kstats.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4r_kstats_gz.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
python3 - <<P
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4r_kstats_gz.csv")))
for r in rows:
    n=r['Name'].split('(')[0].replace('void ','')
    if n.startswith('k_fmt') or n.startswith('k_gz') or 'fill' in n: print(n[:60].ljust(60), r['Calls'], round(float(r["TotalDurationNs"])/1e6/6,2), round(float(r['AverageNs'])/1e6,3))
P
