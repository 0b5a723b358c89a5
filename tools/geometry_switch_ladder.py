"""Sort time of uint32 key + value pairs around the switch from the small geometry (radix_scatter_kernel, 4096-pair tiles) to the
128-byte-line scatter (10 240-pair tiles, one workgroup per CU): the same sizes with either forced (GLU_HIP_SORT_LARGE_MIN).
   python tools/geometry_switch_ladder.py [first n] [last n] [step factor] [pairs|keys|u64|u64keys]      (run twice, once per setting of the switch)"""
import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G
n = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0e6
n_end = float(sys.argv[2]) if len(sys.argv) > 2 else 9.5e6
f = float(sys.argv[3]) if len(sys.argv) > 3 else 1.06
mode = sys.argv[4] if len(sys.argv) > 4 else "pairs"
print("GLU_HIP_SORT_LARGE_MIN =", os.environ.get("GLU_HIP_SORT_LARGE_MIN"), " GLU_HIP_SORT_NT_STORES =", os.environ.get("GLU_HIP_SORT_NT_STORES"))
while n < n_end:
    m = int(n)
    keys = np.random.default_rng(m).integers(0, 2 ** (64 if mode.startswith("u64") else 32), m, dtype=np.uint64 if mode.startswith("u64") else np.uint32)
    vals = np.arange(m, dtype=np.uint32)
    s = G.RadixSort()
    s.prepare_internal_buffers(m, key_bytes=8 if mode.startswith("u64") else 4)
    ts = []
    for r in range(12):
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        if mode == "keys":
            ts.append(G.measure_elapsed_time(lambda: s.sort_keys(kb, m)) * 1e-3)
        elif mode == "u64keys":
            ts.append(G.measure_elapsed_time(lambda: s.sort_keys_ptr(kb.device_ptr(), m, key_bytes=8)) * 1e-3)
        else:
            ts.append(G.measure_elapsed_time(lambda: s(kb, vb, m, 0, key_bytes=8 if mode == "u64" else 4)) * 1e-3)
    ts.sort()
    print("n %9d  min %7.1f us  median %7.1f us" % (m, ts[0], ts[len(ts) // 2]), flush=True)
    n *= f
