"""Sum rocprofv3 counter_collection.csv per (kernel, counter): prints kernel, counter, launches, total, per-launch."""
import csv, sys, collections
tot = collections.defaultdict(float); calls = collections.defaultdict(set)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0].replace("void ", ""); c = r["Counter_Name"]
        tot[(k, c)] += float(r["Counter_Value"]); calls[(k, c)].add(r["Dispatch_Id"])
for (k, c) in sorted(tot):
    n = len(calls[(k, c)])
    print("%-28s %-36s launches %4d total %.6g per_launch %.6g" % (k[:28], c, n, tot[(k, c)], tot[(k, c)] / n))
