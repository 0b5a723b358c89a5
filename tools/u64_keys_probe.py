"""2^28 u64 keys only (no vals): per-pass kernel times, 8-bit vs 4-bit digits (is the 8-bit/4-bit scatter gap of C5 the 64-byte val half-lines?)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G
n = 1 << 28
rng = np.random.default_rng(1)
keys = rng.integers(0, 2**64, n, dtype=np.uint64)
k0 = G.ShaderStorageBuffer(keys)
k = G.ShaderStorageBuffer(size=keys.nbytes)
for bits in (8, 4):
    s = G.RadixSort(digit_bits=bits)
    s.prepare_internal_buffers(n, key_bytes=8, with_vals=False)
    best = 1e18
    for rep in range(4):
        G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), keys.nbytes, 0, 0))
        if rep == 1:
            s.set_profiling(True)
        best = min(best, G.measure_elapsed_time(lambda: s.sort_keys_ptr(k.device_ptr(), n, 0, key_bytes=8)))
    G.synchronize()
    p = s.read_profile()
    passes = max(int(p["passes"]), 1)
    print("keys only %d-bit: %7.3f ms | per pass: count %.3f scan %.3f scatter %.3f ms (%.0f GB/s at 16 B/key)" % (
        bits, best * 1e-6, p["count_ms"] / passes, p["scan_ms"] / passes, p["scatter_ms"] / passes,
        n * 16 / (p["scatter_ms"] / passes * 1e-3) / 1e9), flush=True)
