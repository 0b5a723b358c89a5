"""Launch-bound regime (2^14 .. 2^22 pairs): eager enqueue vs hipGraph replay of the same sort, per-sort device time.
Usage (GPU box): python tools/midsize_probe.py"""
import sys
sys.path.insert(0, "gl-radix-sort_amd")
import torch
import glu_hip as G

REPS = 50
side = torch.cuda.Stream()
sorter = G.RadixSort()
for lg in (13, 14, 16, 18, 20, 21, 22, 24):
    n = 1 << lg
    pk = torch.randint(-(1 << 31), 1 << 31, (n,), device="cuda", dtype=torch.int64).to(torch.int32)
    pv = torch.arange(n, device="cuda", dtype=torch.int32)
    k, v = pk.clone(), pv.clone()
    sorter.prepare_internal_buffers(n)
    torch.cuda.synchronize()

    def one():
        k.copy_(pk)
        v.copy_(pv)
        sorter.run_ptr(k.data_ptr(), v.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)

    def copies():
        k.copy_(pk)
        v.copy_(pv)

    def timed(fn):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / REPS)
        return best * 1e3

    with torch.cuda.stream(side):
        one()
        side.synchronize()
        eager = timed(one)
        eager_c = timed(copies)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            one()
        gc = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gc, stream=side):
            copies()
        g.replay()
        side.synchronize()
        ok = bool((k[1:] ^ -(1 << 31) >= k[:-1] ^ -(1 << 31)).all())
        graph = timed(g.replay)
        graph_c = timed(gc.replay)
    print("2^%d eager %.1f us (copies %.1f)  graph %.1f us (copies %.1f)  sorted=%s" % (lg, eager, eager_c, graph, graph_c, ok), flush=True)
