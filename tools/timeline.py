#!/usr/bin/env python3
"""Per-batch GPU timeline from a rocprofv3 --kernel-trace csv: the kernels of the LAST batch in launch order,
consecutive launches of the same kernel merged, with the idle time in front of each run.  usage: timeline.py <kernel_trace.csv> [first-kernel-of-a-batch]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_sa_lookup<false>"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "")
    m = re.search(r"wrapped_(\w+?)_config<[^,]*, ([\w ]+), ([\w:<> ]+?)>", n)
    if "rocprim" in n:
        kind = "scan" if "scan" in n else "radix" if "radix_sort" in n else "merge" if "merge_sort" in n else "rocprim"
        t = re.findall(r"unsigned long|unsigned int|empty_type|cgx_\w+", n)
        return kind + "<" + ",".join(t[:2]) + ">"
    return n.split("(")[0][:40]
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(first)]
if not starts: sys.exit("no " + first)
bounds = starts + [len(rows)]
seg = next((rows[bounds[i]:bounds[i + 1]] for i in range(len(starts) - 1, -1, -1) if bounds[i + 1] - bounds[i] > 100), rows[starts[-1]:])   # the last full batch
out = []; prev_end = int(seg[0]["Start_Timestamp"]); t0 = prev_end
for r in seg:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    gap = max(0, s - prev_end)
    if out and out[-1][0] == n and gap < 20000: out[-1][1] += 1; out[-1][2] += e - s; out[-1][3] += gap
    else: out.append([n, 1, e - s, gap, s - t0])
    prev_end = max(prev_end, e)
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if short(r["Kernel_Name"]).startswith(first)]
print(f"every launch of {first} in this trace, ms: " + " ".join(f"{d:.4f}" for d in durs) + f"   (median {sorted(durs)[len(durs) // 2]:.4f}; the first launches after the index build touch the tables for the first time)")
busy = sum(o[2] for o in out); idle = sum(o[3] for o in out)
print(f"last batch: {len(seg)} launches, busy {busy/1e6:.1f} ms, idle {idle/1e6:.1f} ms, span {(prev_end-t0)/1e6:.1f} ms")
for n, c, d, g, at in out:
    if d + g >= 100000: print(f"{at/1e6:9.2f} ms  {n:44s} x{c:<4d} {d/1e6:8.2f} ms   idle before {g/1e6:7.2f} ms")
