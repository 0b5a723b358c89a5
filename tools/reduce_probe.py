import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G
n = 1 << 28
d = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint32)
b = G.ShaderStorageBuffer(d)
rd = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)
rd(b, n)
t = min(G.measure_elapsed_time(lambda: rd(b, n)) for _ in range(8)) * 1e-9
print(os.environ.get("GLU_HIP_REDUCE_BLOCKS"), "%.4f ms %.0f GB/s" % (t * 1e3, n * 4 / t / 1e9))
