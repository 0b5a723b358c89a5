"""Diagnostic: the rank worker of tests/test_gpu_dist.py::test_native_multi_rank_sort_over_mock_transport for `world` processes
on one GPU, asynchronous test double, every collective logged, short file time-outs.
usage (GPU box): timeout 300 python tools/mock_async_probe2.py [world] [seg_mode or -] [async 0/1] [rounds]"""
import os, sys, tempfile, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("gl-radix-sort_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seg = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    use_async = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else None
    os.environ.update(GLU_MOCK_RCCL_TIMEOUT_S="15", GLU_MOCK_RCCL_VERBOSE="1")
    import test_gpu_dist as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = tempfile.mkdtemp(prefix="mockprobe")
    uid = os.urandom(128)
    lib = os.path.join(ROOT, "tests/cpp/bin/libmock_rccl.so")
    ps = [ctx.Process(target=T._mock_rank_worker, args=(r, world, uid, lib, d, q, seg, use_async, rounds)) for r in range(world)]
    for p in ps:
        p.start()
    for _ in range(world):
        try:
            rank, out = q.get(timeout=200)
            print(rank, [(o[0], o[1], o[4], o[5], o[6]) for o in out], flush=True)
        except Exception as e:
            print("no result:", repr(e), flush=True)
            break
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    print("exit codes", [p.exitcode for p in ps])
