"""Sorter objects whose value scratch is allocated behind a spacer of S MiB (one candidate placement each:
GLU_HIP_SCRATCH_TUNE_LIST=512:1:S), timed in ONE process on the same caller pairs drawn from 10 consecutive 1 GiB buffers:
a sorter object is fast or slow for every caller pair."""
import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

n = 1 << 28
keys = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint32)
vals = np.arange(n, dtype=np.uint32)
k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
bufs = [G.ShaderStorageBuffer(size=4 * n) for _ in range(10)]
pairs = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (1, 6), (3, 8)]
sorters = []
for spacer in (0, 2048, 4096, 6144, 0, 4096, 1024, 3072):
    os.environ["GLU_HIP_SCRATCH_TUNE_LIST"] = "512:1:%d" % spacer
    q = G.RadixSort(); q.prepare_internal_buffers(n)
    sorters.append((spacer, q))
print("pairs", pairs)
for spacer, q in sorters:
    row = []
    for (i, j) in pairs:
        t = 1e9
        for _ in range(2):
            G.check(G.lib().glu_buffer_copy(k0.handle(), bufs[i].handle(), 4 * n, 0, 0)); G.check(G.lib().glu_buffer_copy(v0.handle(), bufs[j].handle(), 4 * n, 0, 0))
            t = min(t, G.measure_elapsed_time(lambda: q(bufs[i], bufs[j], n)) * 1e-6)
        row.append("%.3f" % t)
    print("scratch spacer %5d MiB: %s   mean %.3f" % (spacer, " ".join(row), sum(map(float, row)) / len(row)), flush=True)
