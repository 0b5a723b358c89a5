"""Is the speed lottery of large arrays a property of each ARRAY (where it lies) or of PAIRS of arrays (how they lie to each
other)?  B buffers of 1 GiB in one process: read rate of each (Reduce), write rate of each (fill), copy rate of every
ordered pair, and the 2^28-pair sort on every (keys, vals) choice among them.   python tools/placement_probe.py [buffers]"""
import sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 1 << 28
src = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint32)
keep = G.ShaderStorageBuffer(src)
bufs = [G.ShaderStorageBuffer(size=4 * n) for _ in range(B)]
rd = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)
best = lambda f, r=4: min(G.measure_elapsed_time(f) for _ in range(r)) * 1e-6
print("device pointers:", " ".join("%x" % b.device_ptr() for b in bufs))
for i, b in enumerate(bufs):
    G.check(G.lib().glu_buffer_copy(keep.handle(), b.handle(), 4 * n, 0, 0))
    t_fill = best(lambda: b.clear(7))
    G.check(G.lib().glu_buffer_copy(keep.handle(), b.handle(), 4 * n, 0, 0))
    print("buffer %d: fill %.3f ms (%.0f GB/s)" % (i, t_fill, 4 * n / t_fill / 1e6), flush=True)
print("copy i -> j (ms):")
for i in range(B):
    row = []
    for j in range(B):
        if i == j:
            row.append("   -  ")
            continue
        row.append("%6.3f" % best(lambda: G.check(G.lib().glu_buffer_copy(bufs[i].handle(), bufs[j].handle(), 4 * n, 0, 0))))
    print("  from %d: %s" % (i, " ".join(row)), flush=True)
s = G.RadixSort(); s.prepare_internal_buffers(n)
vals = np.arange(n, dtype=np.uint32)
keepv = G.ShaderStorageBuffer(vals)
print("sort with keys in buffer i, values in buffer j (ms):")
for i in range(B):
    row = []
    for j in range(B):
        if i == j:
            row.append("   -  ")
            continue
        t = 1e9
        for _ in range(2):
            G.check(G.lib().glu_buffer_copy(keep.handle(), bufs[i].handle(), 4 * n, 0, 0))
            G.check(G.lib().glu_buffer_copy(keepv.handle(), bufs[j].handle(), 4 * n, 0, 0))
            t = min(t, G.measure_elapsed_time(lambda: s(bufs[i], bufs[j], n)) * 1e-6)
        row.append("%6.3f" % t)
    print("  keys %d: %s" % (i, " ".join(row)), flush=True)
S = 4
sorters = []
for _ in range(S):
    q = G.RadixSort(); q.prepare_internal_buffers(n); sorters.append(q)
pairs = [(0, 1), (2, 3), (4, 5), (0, 3), (1, 4)][: max(1, B // 2 + 2)]
print("sort time (ms) by sorter object (its scratch arrays) and caller pair; last column: copy rate between the sorter's own scratch arrays is not visible from here")
for qi, q in enumerate(sorters):
    row = []
    for (i, j) in pairs:
        t = 1e9
        for _ in range(2):
            G.check(G.lib().glu_buffer_copy(keep.handle(), bufs[i].handle(), 4 * n, 0, 0))
            G.check(G.lib().glu_buffer_copy(keepv.handle(), bufs[j].handle(), 4 * n, 0, 0))
            t = min(t, G.measure_elapsed_time(lambda: q(bufs[i], bufs[j], n)) * 1e-6)
        row.append("%6.3f" % t)
    print("  sorter %d: %s   (pairs %s)" % (qi, " ".join(row), pairs), flush=True)
