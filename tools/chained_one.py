import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
keys = np.random.default_rng(m).integers(0, 2 ** 32, m, dtype=np.uint32)
vals = np.arange(m, dtype=np.uint32)
os.environ["GLU_HIP_SORT_CHAINED"] = "1"
s = G.RadixSort()
s.prepare_internal_buffers(m)
for r in range(6):
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    s(kb, vb, m)
    G.synchronize()
