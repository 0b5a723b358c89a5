"""Round 4: from which size does the sort that ends in LDS pay?  An object with the attempt switched off (GLU_HIP_SORT_LDS_FINISH=0)
against one with the library's defaults -- or, with "any", one that pairs its passes and makes the attempt from any size
(GLU_HIP_SORT_PAIR_MIN=1, GLU_HIP_SORT_FINISH_MIN=1) -- 2^22 .. 2^26 pairs of 32-bit (or, with "u64", 64-bit) keys.
   python tools/finish_midsize_probe.py [any] [u64]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gl-radix-sort_amd"))
import glu_hip as G
import torch

ANY = "any" in sys.argv
KB = 8 if "u64" in sys.argv else 4


def sorter(attempt):
    env = ({"GLU_HIP_SORT_PAIR_MIN": "1", "GLU_HIP_SORT_FINISH_MIN": "1"} if ANY else {}) if attempt else {"GLU_HIP_SORT_LDS_FINISH": "0"}
    os.environ.update(env)
    try:
        return G.RadixSort()
    finally:
        for k in env:
            del os.environ[k]


def timed(s, k0, v0, n):
    k, v = k0.clone(), v0.clone()
    ms = []
    for _ in range(12):
        k.copy_(k0); v.copy_(v0)
        torch.cuda.synchronize(); G.synchronize()
        t0 = time.perf_counter()
        s.run_ptr(k.data_ptr(), v.data_ptr(), n, key_bytes=KB)
        G.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return k, sorted(ms[2:])


for lg2 in (22, 23, 24, 24.5, 25, 25.25, 25.5, 25.75, 26):
    n = int(2 ** lg2) + 4 * 77
    if KB == 4:
        k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0")
    else:
        k0 = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda:0")
    v0 = torch.arange(n, dtype=torch.int32, device="cuda:0")
    a, b = sorter(False), sorter(True)
    a.prepare_internal_buffers(n, key_bytes=KB); b.prepare_internal_buffers(n, key_bytes=KB)
    ka, ta = timed(a, k0, v0, n)
    kb, tb = timed(b, k0, v0, n)
    print("2^%.2f pairs, %d-byte keys: attempt off median %.3f min %.3f ms | %s median %.3f min %.3f ms  %s  same keys: %s" % (
        lg2, KB, ta[len(ta) // 2], ta[0], "attempt from any size" if ANY else "library defaults", tb[len(tb) // 2], tb[0],
        b.read_finish(), bool((ka == kb).all())), flush=True)
