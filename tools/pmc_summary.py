"""Summarise rocprofv3 --pmc CSV output: per kernel name, average counter value per dispatch.
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
    print(k[:150])
    for c, v in sorted(cs.items()):
        print("   %-24s n=%3d avg=%.6g" % (c, len(v), sum(v) / len(v)))
