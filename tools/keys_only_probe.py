"""Round 4: keys-only sort of 2^28 uint32 keys with and without the attempt to end in LDS (12 B/key/pass against 8 + 8)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gl-radix-sort_amd"))
import glu_hip as G
import torch

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0")
for finish in (0, 1, 0, 1):
    os.environ["GLU_HIP_SORT_LDS_FINISH"] = str(finish)
    s = G.RadixSort()
    s.prepare_internal_buffers(n, with_vals=False)
    k = k0.clone()
    ms = []
    for rep in range(8):
        k.copy_(k0)
        torch.cuda.synchronize(); G.synchronize()
        t0 = time.perf_counter()
        s.sort_keys_ptr(k.data_ptr(), n)
        G.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    f = k ^ torch.tensor(-2**31, dtype=torch.int32, device="cuda:0")
    print("keys only, 2^%d keys, attempt %s: median %.3f min %.3f ms  %s  sorted: %s" % (
        n.bit_length() - 1, "on" if finish else "off", sorted(ms[1:])[len(ms) // 2 - 1], min(ms[1:]), s.read_finish(), bool((f[1:] >= f[:-1]).all())), flush=True)
    s.destroy()
