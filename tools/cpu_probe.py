import ctypes, numpy as np, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libglu_cpu_baseline.so"))
L.glu_cpu_sort_pairs.restype = ctypes.c_double
L.glu_cpu_sort_pairs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
print("hw threads", L.glu_cpu_hardware_threads(), "affinity", len(os.sched_getaffinity(0)))
for logn in (25, 27):
    n = 1 << logn
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32); vals = np.arange(n, dtype=np.uint32)
    for th in (1, 8, 16, 32, 64, 128, 256):
        k, v = keys.copy(), vals.copy()
        t = L.glu_cpu_sort_pairs(k.ctypes.data, v.ctypes.data, n, th)
        print("2^%d threads %3d: %.3f s  %.1f Mkeys/s" % (logn, th, t, n / t / 1e6), flush=True)
