import os, sys, tempfile, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("gl-radix-sort_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
if __name__ == "__main__":
    import test_gpu_dist as T
    for use_async in (False, True):
        ctx = mp.get_context("spawn"); q = ctx.Queue(); d = tempfile.mkdtemp(prefix="fault"); uid = os.urandom(128)
        lib = os.path.join(ROOT, "tests/cpp/bin/libmock_rccl.so")
        ps = [ctx.Process(target=T._mock_fault_worker, args=(r, 2, uid, lib, d, q, "no_hist_wait", use_async)) for r in range(2)]
        [p.start() for p in ps]
        res = dict(q.get(timeout=120) for _ in range(2))
        [p.join(timeout=20) for p in ps]
        print("async" if use_async else "sync", {r: [(x[0], x[1]) if x[0] != "error" else x for x in v] for r, v in res.items()}, flush=True)
