"""Round 4: the sort that ends in LDS against the four ordinary passes -- same results, sort times.
   python tools/finish_probe.py [log2 sizes ...]"""
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gl-radix-sort_amd"))
import glu_hip as G
import torch


def sorter(finish):
    os.environ["GLU_HIP_SORT_LDS_FINISH"] = "1" if finish else "0"
    try:
        return G.RadixSort()
    finally:
        del os.environ["GLU_HIP_SORT_LDS_FINISH"]


def timed(srt, k0, v0, n, reps=7):
    k, v = k0.clone(), v0.clone()
    ms = []
    for _ in range(reps):
        k.copy_(k0); v.copy_(v0)
        torch.cuda.synchronize()
        G.synchronize()
        t0 = time.perf_counter()
        srt.run_ptr(k.data_ptr(), v.data_ptr(), n)
        G.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return k, v, sorted(ms)


def main():
    # arguments: log2 sizes; "b31" / "b30" before a size: keys of that many bits (the top bits zero) from there on
    dev = torch.device("cuda:0")
    cases, bits = [], 32
    for a in sys.argv[1:] or ["26", "27", "28"]:
        if a.startswith("b"):
            bits = int(a[1:])
        else:
            cases.append((int(a), bits))
    for lg, bits in cases:
        for n in ((1 << lg), (1 << lg) - 12345):
            g = torch.Generator(device=dev); g.manual_seed(lg)
            k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device=dev, generator=g)
            if bits < 32:
                k0 = (k0 >> (32 - bits)) & torch.tensor((1 << bits) - 1, dtype=torch.int32, device=dev)
            v0 = torch.arange(n, dtype=torch.int32, device=dev)
            a, b = sorter(False), sorter(True)
            for s in (a, b):
                s.prepare_internal_buffers(n)
            ka, va, ta = timed(a, k0, v0, n)
            kb, vb, tb = timed(b, k0, v0, n)
            same = bool((ka == kb).all()) and bool((va == vb).all())
            print("2^%d%s pairs%s: ordinary median %.3f min %.3f ms | LDS finish median %.3f min %.3f ms  %s  same result: %s" % (
                lg, "" if n == 1 << lg else "-12345", "" if bits == 32 else " of %d-bit keys" % bits, ta[len(ta) // 2], ta[0], tb[len(tb) // 2], tb[0], b.read_finish(), same), flush=True)
            if not same:
                bad = (ka != kb).nonzero()
                print("  first differing positions", bad[:8].flatten().tolist(), "of", int(bad.numel()))
                sys.exit(1)
            del a, b


main()
