"""Inclusive... exclusive scan timing ladder: chained single-pass vs reduce-then-scan (GLU_HIP_SCAN_CHAINED=0).
Usage (GPU box): python tools/scan_probe.py"""
import os, subprocess, sys

CHILD = r"""
import sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G
for lg in (14, 16, 17, 18, 19, 20, 21, 22, 24, 26, 28):
    n = 1 << lg
    d = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    sc = G.BlellochScan(G.DataType_Uint)
    sc(b, n)
    t = min(G.measure_elapsed_time(lambda: sc(b, n)) for _ in range(12)) * 1e-9
    print("2^%d %.4f ms %.0f GB/s" % (lg, t * 1e3, n * 8 / t / 1e9), flush=True)
"""
for mode in ("1", "0"):
    print("GLU_HIP_SCAN_CHAINED=" + mode, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, GLU_HIP_SCAN_CHAINED=mode), check=False)
