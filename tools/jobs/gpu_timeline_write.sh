#!/bin/bash
# kernel trace of real steps (files written): every launch of the north-star kernel with its duration, and what overlapped it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-tlw}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tlw_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_run.log 2>&1; echo "rc=$?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tlw_tmp -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sa_launches.txt <<Q
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    if "k_sa_lookup<false>" in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        over = [x["Kernel_Name"].split("(")[0][:30] for x in rows[max(0, i - 40):i + 40] if x is not r and int(x["Start_Timestamp"]) < e and int(x["End_Timestamp"]) > s]
        print(f"k_sa_lookup<false> {(e - s) / 1e6:.4f} ms, queue {r.get('Queue_Id')}, overlapping kernels: {sorted(set(over))}")
Q
ls $GRAFT_REPO_ROOT/gpurun_out/tlw_tmp/*/ | head; rm -rf $GRAFT_REPO_ROOT/gpurun_out/tlw_tmp; cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sa_launches.txt
