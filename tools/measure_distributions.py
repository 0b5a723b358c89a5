"""Sort time vs key distribution (2^26 pairs): looks for pathological slow-downs (LDS atomic conflicts in the count
kernel, degenerate runs in the scatter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

n = 1 << 26
rng = np.random.default_rng(3)
dists = {
    "uniform": rng.integers(0, 2**32, n, dtype=np.uint32),
    "all zero": np.zeros(n, dtype=np.uint32),
    "2 values": (rng.integers(0, 2, n, dtype=np.uint32) * np.uint32(0x01010101)),
    "3 values/byte": (rng.integers(0, 3, n, dtype=np.uint32) * np.uint32(0x55555555)),
    "10 values": rng.integers(0, 10, n, dtype=np.uint32) * np.uint32(0x10305070),
    "sorted": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32)),
    "reverse sorted": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32))[::-1].copy(),
    "low 8 bits only": rng.integers(0, 256, n, dtype=np.uint32),
    "alternating 0/max": np.where(np.arange(n) % 2 == 0, 0, 0xFFFFFFFF).astype(np.uint32),
    "runs of 64 equal": np.repeat(rng.integers(0, 2**32, n // 64, dtype=np.uint32), 64),
    "gaussian-ish": (rng.normal(2**31, 2**27, n).clip(0, 2**32 - 1)).astype(np.uint32),
}
vals = np.arange(n, dtype=np.uint32)
v0 = G.ShaderStorageBuffer(vals)
for bits in (8, 4):
    s = G.RadixSort(digit_bits=bits); s.prepare_internal_buffers(n); s.set_profiling(True)
    for name, keys in dists.items():
        k0 = G.ShaderStorageBuffer(keys)
        k, v = G.ShaderStorageBuffer(size=4 * n), G.ShaderStorageBuffer(size=4 * n)
        best = 1e18
        for _ in range(3):
            G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), 4 * n, 0, 0))
            G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), 4 * n, 0, 0))
            s.read_profile()
            t = G.measure_elapsed_time(lambda: s(k, v, n))
            p = s.read_profile()
            if t < best:
                best, bp = t, p
        print("bits %d %-18s %7.3f ms  %8.1f Mkeys/s   count %.3f scatter %.3f ms/pass" % (
            bits, name, best * 1e-6, n / best * 1e3, bp["count_ms"] / bp["passes"], bp["scatter_ms"] / bp["passes"]), flush=True)
