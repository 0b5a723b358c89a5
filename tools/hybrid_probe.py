"""Experiment: MSD first pass (top 8 bits) + cache-resident LSD passes on bucket groups, vs the plain 4 x 8-bit LSD sort.
Host-driven (reads the bucket histogram back), not a product path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np, torch
import glu_hip as G

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
dev = torch.device("cuda:0")
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
h = st.cuda_stream
g = torch.Generator(device=dev); g.manual_seed(1)
keys0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device=dev, generator=g)
vals0 = torch.arange(n, dtype=torch.int32, device=dev)
s = G.RadixSort(); s.prepare_internal_buffers(n)

def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3

k = keys0.clone(); v = vals0.clone()
def plain():
    k.copy_(keys0); v.copy_(vals0)
t_copy = timed(plain)
def plain_sort():
    k.copy_(keys0); v.copy_(vals0); s.run_ptr(k.data_ptr(), v.data_ptr(), n, 0, h)
print("restore copy %.3f ms; plain 4x8-bit sort %.3f ms" % (t_copy, timed(plain_sort) - t_copy), flush=True)
ref_k = k.clone(); ref_v = v.clone()

k2 = torch.empty_like(k); v2 = torch.empty_like(v)
hist = torch.zeros(256, dtype=torch.int32, device=dev)
def hybrid(group_buckets, local_bits):
    # MSD pass: k -> k2
    s.partition_ptr(keys0.data_ptr(), vals0.data_ptr(), k2.data_ptr(), v2.data_ptr(), n, 24, 8, hist.data_ptr(), h)
    hh = hist.cpu().numpy().astype(np.int64)
    off = np.concatenate([[0], np.cumsum(hh)])
    shifts = []
    sh = 0
    for b in local_bits:
        shifts.append((sh, b)); sh += b
    assert sh == 24 and len(shifts) % 2 == 0
    for g0 in range(0, 256, group_buckets):
        lo, hi = int(off[g0]), int(off[min(g0 + group_buckets, 256)])
        cnt = hi - lo
        if cnt == 0: continue
        a_k, a_v, b_k, b_v = k2.data_ptr() + 4 * lo, v2.data_ptr() + 4 * lo, k.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo
        for (shift, bits) in shifts:
            s.partition_ptr(a_k, a_v, b_k, b_v, cnt, shift, bits, None, h)
            a_k, b_k = b_k, a_k; a_v, b_v = b_v, a_v
    # result in k2/v2
for local_bits in ([8, 8, 4, 4], [6, 6, 6, 6], [8, 8, 8]):
    if len(local_bits) % 2: continue
    for gb in (1, 2, 4, 8, 16, 32, 64, 256):
        t = timed(lambda: hybrid(gb, local_bits), reps=2)
        ok = bool((k2 == ref_k).all()) and bool((v2 == ref_v).all())
        print("hybrid local bits %s group %3d buckets (%6.1f MiB of pairs): %.3f ms %s" % (local_bits, gb, gb * n / 256 * 8 / 2**20, t, "ok" if ok else "WRONG"), flush=True)
