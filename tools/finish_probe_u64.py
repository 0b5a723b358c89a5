"""Round 4: 64-bit keys + 32-bit values, the sort that ends in LDS (two top-bit passes + six in-LDS rounds on the low 48 bits)
against the eight ordinary passes -- same results, sort times.   python tools/finish_probe_u64.py [log2 sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gl-radix-sort_amd"))
import glu_hip as G
import torch


def sorter(finish):
    os.environ["GLU_HIP_SORT_LDS_FINISH"] = "1" if finish else "0"
    try:
        return G.RadixSort()
    finally:
        del os.environ["GLU_HIP_SORT_LDS_FINISH"]


def timed(srt, k0, v0, n, reps=6):
    k, v = k0.clone(), v0.clone()
    ms = []
    for _ in range(reps):
        k.copy_(k0); v.copy_(v0)
        torch.cuda.synchronize(); G.synchronize()
        t0 = time.perf_counter()
        srt.run_ptr(k.data_ptr(), v.data_ptr(), n, key_bytes=8)
        G.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return k, v, sorted(ms[1:])


dev = torch.device("cuda:0")
for lg in [int(a) for a in sys.argv[1:]] or [26, 27, 28]:
    n = (1 << lg) - (0 if lg == 28 else 4321)
    k0 = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev)
    v0 = torch.arange(n, dtype=torch.int32, device=dev)
    a, b = sorter(False), sorter(True)
    for s in (a, b):
        s.prepare_internal_buffers(n, key_bytes=8)
    ka, va, ta = timed(a, k0, v0, n)
    kb, vb, tb = timed(b, k0, v0, n)
    same = bool((ka == kb).all()) and bool((va == vb).all())
    print("%d u64 + u32 pairs: eight passes median %.3f min %.3f ms | ending in LDS median %.3f min %.3f ms  %s  same result: %s" % (
        n, ta[len(ta) // 2], ta[0], tb[len(tb) // 2], tb[0], b.read_finish(), same), flush=True)
    if not same:
        sys.exit(1)
    a.destroy(); b.destroy()
