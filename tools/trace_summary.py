"""Per-sort kernel durations from a rocprofv3 --kernel-trace CSV: for every sort (ended by radix_finalize_kernel) the durations
(us) of its scatter launches that moved data (S), of the in-LDS pass (F), and of everything else summed (o).
   python tools/trace_summary.py <kernel_trace.csv> [last N sorts]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sorts, cur, other, t0 = [], [], 0.0, None
for r in rows:
    n = r["Kernel_Name"]
    if "glu_hip::" not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if t0 is None:
        t0 = int(r["Start_Timestamp"])
    if "radix_finalize" in n:
        sorts.append((cur, round(other), round((int(r["End_Timestamp"]) - t0) / 1e3)))
        cur, other, t0 = [], 0.0, None
    elif "radix_scatter_lines" in n and d > 100:
        cur.append("S%d" % round(d))
    elif "radix_finish_sort" in n:
        cur.append("F%d" % round(d))
    elif "pair_count" in n and d > 100:
        cur.append("C%d" % round(d))
    else:
        other += d
for cur, other, span in sorts[-last:]:
    print(" ".join(cur), " other kernels %d us, first start to last end %d us" % (other, span))
