import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np, torch
import glu_hip as G
for log2n in (24, 26, 27, 28):
    n = 1 << log2n
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ok, ov = torch.empty_like(keys), torch.empty_like(vals)
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    s = G.RadixSort()
    s.partition_ptr(keys.data_ptr(), vals.data_ptr(), ok.data_ptr(), ov.data_ptr(), n, 24, 8, hist.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = hist.cpu().numpy().astype(np.int64)
    ref = torch.bincount(((keys >> 24) & 0xFF).to(torch.int64), minlength=256).cpu().numpy()
    print(log2n, h.sum(), n, (h == ref).all(), h[:4], ref[:4])
