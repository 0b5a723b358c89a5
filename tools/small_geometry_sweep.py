"""Sort time over the launch-bound sizes for the library named by GLU_HIP_LIB_PATH (tuning builds of the small geometry:
-DGLU_SMALL_THREADS / _KPT / _BLOCKS_PER_CU): python tools/small_geometry_sweep.py [pairs|keys] [digit bits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np, glu_hip as G
mode = sys.argv[1] if len(sys.argv) > 1 else "pairs"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = []
for n in ([1 << 20, 1 << 21, 2600000, 3000000, 3300000, 3600000, 3900000] if os.environ.get("SWEEP_HIGH") else [16385, 24000, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 3000000, 3900000]):
    lg = np.log2(n)
    keys = np.random.default_rng(n).integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    s = G.RadixSort(digit_bits=bits)
    s.prepare_internal_buffers(n)
    k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    kb, vb = G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)
    best = 1e18
    for r in range(30):
        G.check(G.lib().glu_buffer_copy(k0.handle(), kb.handle(), keys.nbytes, 0, 0))
        G.check(G.lib().glu_buffer_copy(v0.handle(), vb.handle(), vals.nbytes, 0, 0))
        best = min(best, G.measure_elapsed_time((lambda: s.sort_keys(kb, n)) if mode == "keys" else (lambda: s(kb, vb, n))))
    ok = (kb.get_data(np.uint32) == np.sort(keys)).all()
    out.append("2^%.1f %5.1f%s" % (lg, best * 1e-3, "" if ok else " WRONG"))
print("%-22s %s %d-bit" % (os.path.basename(os.environ.get("GLU_HIP_LIB_PATH", "default")), mode, bits), " | ".join(out), flush=True)
