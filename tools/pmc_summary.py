"""Summarise rocprofv3 --pmc CSV output: per kernel name, average counter value per dispatch.
Dispatches whose value is below 1 % of the kernel's largest are left out of the average and counted separately: a sort that
tries to end in LDS enqueues two sequences of passes and the kernels of the one not taken return at once (radix_lds_finish.hpp).
usage: python tools/pmc_summary.py <dir> [substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub and sub not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        top = max(v)
        kept = [x for x in v if x >= 0.01 * top] if top > 0 else v
        note = "" if len(kept) == len(v) else "   (+ %d dispatches that returned at once, not averaged)" % (len(v) - len(kept))
        print("   %-24s n=%3d avg=%.6g%s" % (c, len(kept), sum(kept) / len(kept), note))
