"""A few sorts of 2^k pairs back to back (for rocprofv3 --kernel-trace: per-kernel durations and the gaps between them).
python tools/midsize_trace.py [log2n] ; then tools/midsize_trace.py --analyze <kernel_trace.csv>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "glu_hip" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last complete sort: from the last count kernel with shift... simply the last 13 kernels
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    last = rows[-per:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) if prev_end is not None else 0
        print("%-60s start %7.2f us  dur %6.2f us  gap before %5.2f us" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3))
        prev_end = e
    print("total %.2f us" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np, glu_hip as G
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
keys = np.random.default_rng(1).integers(0, 2**32, n, dtype=np.uint32)
vals = np.arange(n, dtype=np.uint32)
s = G.RadixSort()
s.prepare_internal_buffers(n)
for rep in range(6):
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    t = G.measure_elapsed_time(lambda: s(kb, vb, n))
print("2^%d pairs: %.1f us" % (log2n, t * 1e-3))
