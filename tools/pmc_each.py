"""Per-dispatch values of a rocprofv3 --pmc CSV directory, in dispatch order, for kernels whose name contains a substring.
usage: python tools/pmc_each.py <dir> <substring>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row.get("Kernel_Name", ""):
            rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"]), row["Kernel_Name"][:70]))
for d, c, v, k in sorted(rows):
    print("%6d %-12s %12.0f  %s" % (d, c, v, k))
