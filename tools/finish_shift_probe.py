"""Is a counting pass on key bits [16, 24) of uniform keys slower than one on bits [0, 8)?  (round 4: the first top-bit pass of the
sort that ends in LDS takes 0.96-1.0 ms where the first pass of the ordinary sort takes 0.87-0.92)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gl-radix-sort_amd"))
import glu_hip as G
import torch

n = 1 << 28
dev = torch.device("cuda:0")
k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device=dev)
v0 = torch.arange(n, dtype=torch.int32, device=dev)
os.environ["GLU_HIP_SORT_LDS_FINISH"] = "0"
s = G.RadixSort()
s.prepare_internal_buffers(n)
k, v = k0.clone(), v0.clone()
for lo, hi in [(0, 16), (16, 32), (0, 16), (16, 32), (8, 24)]:
    s.set_profiling(False)
    for rep in range(6):
        if rep == 1:
            s.set_profiling(True)
        k.copy_(k0); v.copy_(v0)
        torch.cuda.synchronize()
        s.sort_bit_range_ptr(k.data_ptr(), v.data_ptr(), n, lo, hi)
        G.synchronize()
    p = s.read_profile()
    print("bits [%2d, %2d): %d passes  count %.3f  scan %.3f  scatter %.3f ms per pass" % (
        lo, hi, p["passes"], p["count_ms"] / p["passes"], p["scan_ms"] / p["passes"], p["scatter_ms"] / p["passes"]), flush=True)
