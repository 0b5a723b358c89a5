"""Mid-size sorts, 2^14 .. 2^22 pairs: the per-pass three-launch form (count, row scan, scatter) against the chained form
(GLU_HIP_SORT_CHAINED=1: one histogram launch + one look-back scatter launch per pass), same box, same inputs, device time
of the sort alone (best of 12), results compared with each other and with numpy.
  python tools/chained_ladder.py"""
import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

print(G.device_info())
sizes = []
for lg in range(14, 23):
    sizes += [1 << lg, (1 << lg) + (1 << (lg - 1))]
for m in sizes:
    if m > (1 << 22):
        break
    keys = np.random.default_rng(m).integers(0, 2 ** 32, m, dtype=np.uint32)
    vals = np.arange(m, dtype=np.uint32)
    row, outs = [], []
    for chained in ("0", "1"):
        os.environ["GLU_HIP_SORT_CHAINED"] = chained
        s = G.RadixSort()
        s.prepare_internal_buffers(m)
        best = 1e18
        for r in range(12):
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            best = min(best, G.measure_elapsed_time(lambda: s(kb, vb, m)))
        row.append(best * 1e-3)
        outs.append((kb.get_data(np.uint32), vb.get_data(np.uint32)))
    order = np.argsort(keys, kind="stable")
    ok = (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all() and (outs[1][0] == keys[order]).all() and (outs[1][1] == vals[order]).all()
    print("n %8d (2^%5.2f): three launches per pass %7.1f us   chained %7.1f us   %+5.1f %%   %s" % (
        m, np.log2(m), row[0], row[1], (row[1] / row[0] - 1) * 100, "ok" if ok else "WRONG"), flush=True)
