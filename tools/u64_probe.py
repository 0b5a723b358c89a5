"""C5 (2^28 u64 keys + u32 vals): sort time and per-kernel averages, line kernel vs element-granular kernel.
python tools/u64_probe.py [log2n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
rng = np.random.default_rng(1)
keys = rng.integers(0, 2**64, n, dtype=np.uint64)
vals = np.arange(n, dtype=np.uint32)
k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
k, v = G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)
for label, env in (("lines", None),):
    if env:
        os.environ["GLU_HIP_SORT_NO_LINES"] = env
    else:
        os.environ.pop("GLU_HIP_SORT_NO_LINES", None)
    for bits in (8, 4):
        s = G.RadixSort(digit_bits=bits)
        s.prepare_internal_buffers(n, key_bytes=8)
        best = 1e18
        for rep in range(4):
            G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), keys.nbytes, 0, 0))
            G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), vals.nbytes, 0, 0))
            if rep == 1:
                s.set_profiling(True)
            best = min(best, G.measure_elapsed_time(lambda: s(k, v, n, 0, key_bytes=8)))
        G.synchronize()
        p = s.read_profile()
        passes = max(int(p["passes"]), 1)
        print("%-17s %d-bit digits: %7.3f ms | per pass: count %.3f scan %.3f scatter %.3f ms (%.0f GB/s at 24 B/pair)" % (
            label, bits, best * 1e-6, p["count_ms"] / passes, p["scan_ms"] / passes, p["scatter_ms"] / passes,
            n * 24 / (p["scatter_ms"] / passes * 1e-3) / 1e9), flush=True)
