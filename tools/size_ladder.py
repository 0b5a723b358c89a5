"""Sort time over a fine size ladder (x1.25 steps): a quick way to spot cliffs at the switch points between the
one-workgroup path, the fused-scan path, the small and the large geometry and the pass plan.
usage (GPU box): python tools/size_ladder.py [keys|pairs|u64] [first n] [last n]"""
import sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

mode = sys.argv[1] if len(sys.argv) > 1 else "pairs"
n = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
n_end = float(sys.argv[3]) if len(sys.argv) > 3 else float(1 << 27)
prev = None
while n < n_end:
    m = int(n)
    dt = np.uint64 if mode == "u64" else np.uint32
    keys = np.random.default_rng(m).integers(0, 2 ** (64 if mode == "u64" else 32), m, dtype=dt)
    vals = np.arange(m, dtype=np.uint32)
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
    us = best * 1e-3
    jump = "  <-- +%.0f %% for +25 %% elements" % ((us / prev - 1) * 100) if prev is not None and us > prev * 1.45 else ""
    print("n %10d: %9.1f us  %9.1f Mkeys/s%s" % (m, us, m / us, jump), flush=True)
    prev = us
    n *= 1.25
