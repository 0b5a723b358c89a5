"""Sort time over sizes for two settings of GLU_HIP_SORT_LARGE_MIN (where the line kernel takes over from the small
geometry): python tools/ladder_ab.py [pairs|keys|u64]"""
import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

mode = sys.argv[1] if len(sys.argv) > 1 else "pairs"
n = 20000.0
while n < 6.0e6:
    m = int(n)
    dt = np.uint64 if mode == "u64" else np.uint32
    keys = np.random.default_rng(m).integers(0, 2 ** (64 if mode == "u64" else 32), m, dtype=dt)
    vals = np.arange(m, dtype=np.uint32)
    row = []
    for large_min in (None, "1"):
        if large_min:
            os.environ["GLU_HIP_SORT_LARGE_MIN"] = large_min
        else:
            os.environ.pop("GLU_HIP_SORT_LARGE_MIN", None)
        s = G.RadixSort()
        s.prepare_internal_buffers(m, key_bytes=8 if mode == "u64" else 4)
        best = 1e18
        for r in range(8):
            kb = G.ShaderStorageBuffer(keys)
            if mode == "keys":
                best = min(best, G.measure_elapsed_time(lambda: s.sort_keys(kb, m)))
            else:
                vb = G.ShaderStorageBuffer(vals)
                best = min(best, G.measure_elapsed_time(lambda: s(kb, vb, m, 0, key_bytes=8 if mode == "u64" else 4)))
        row.append(best * 1e-3)
    print("n %9d: default %8.1f us   line kernel from one tile up %8.1f us   %s" % (m, row[0], row[1], "<-- lines faster" if row[1] < row[0] * 0.97 else ""), flush=True)
    n *= 1.3
