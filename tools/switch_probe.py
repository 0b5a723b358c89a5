import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gl-radix-sort_amd"))
import numpy as np, glu_hip as G
for n in (3_500_000, 4_000_000, 4_500_000, 5_000_000, 5_500_000, 6_000_000, 7_000_000, 8_000_000, 10_000_000, 13_000_000):
    keys = np.random.default_rng(n).integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    kb, vb = G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)
    row = []
    for large_min in (None, "100000000"):
        if large_min:
            os.environ["GLU_HIP_SORT_LARGE_MIN"] = large_min
        else:
            os.environ.pop("GLU_HIP_SORT_LARGE_MIN", None)
        s = G.RadixSort()
        s.prepare_internal_buffers(n)
        best = 1e18
        for r in range(12):
            G.check(G.lib().glu_buffer_copy(k0.handle(), kb.handle(), keys.nbytes, 0, 0))
            G.check(G.lib().glu_buffer_copy(v0.handle(), vb.handle(), vals.nbytes, 0, 0))
            best = min(best, G.measure_elapsed_time(lambda: s(kb, vb, n)))
        row.append(best * 1e-3)
    print("n %9d: default %7.1f us   small geometry %7.1f us" % (n, row[0], row[1]), flush=True)
