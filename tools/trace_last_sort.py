"""The kernels of the LAST sort in a rocprofv3 --kernel-trace CSV, one line each: start offset (us) from the sort's first kernel,
duration (us), name (a sort = everything after the last-but-one buffer-copy pair of tools/sort_loop.py).
   python tools/trace_last_sort.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(i for i, r in enumerate(rows) if "glu_hip::" in r["Kernel_Name"])          # the last kernel of the last sort
begin = max([i for i in range(end) if "copy" in rows[i]["Kernel_Name"].lower()] + [-1]) + 1  # behind the copy that restored its input
seq = rows[begin:end + 1]  # (kernels in between that are not the library's -- fills -- are listed too)
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("glu_hip::", "").replace("void ", ""))[:110]
    print("%9.1f  gap %6.1f  dur %8.1f  grid %8s wg %5s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), name))
    prev_end = e
print("kernels %d, first start to last end %.1f us, sum of durations %.1f us" % (len(seq), (prev_end - t0) / 1e3, sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq)))
