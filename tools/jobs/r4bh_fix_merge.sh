#!/bin/bash
# round 4: the fix pass of run_sort as a merge of the two sorted pieces: parity, then per-kernel times (no files)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q > gpurun_out/r4bh_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r4bh_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-write > $GRAFT_REPO_ROOT/gpurun_out/r4bh_kstats.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4bh_kstats_nowrite.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
python3 - <<P
import csv, json
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4bh_kstats_nowrite.csv")))
for r in rows[:26]:
    print(r['Name'].split('(')[0].replace('void ','')[:70].ljust(70), r['Calls'], round(float(r["TotalDurationNs"])/1e6/8,2), round(float(r['AverageNs'])/1e6,3))
j=json.loads([l for l in open("$GRAFT_REPO_ROOT/gpurun_out/r4bh_kstats.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]; print("under profiler:", j["ms_per_step"], {k:s[k] for k in ("gappy","extract","lexicon")})
P
