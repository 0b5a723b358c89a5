"""Diagnostic: two processes on one GPU as ranks of glu_dist over the test double in ASYNC mode, a handful of sorts, every
collective logged (GLU_MOCK_RCCL_VERBOSE).  usage (GPU box): timeout 200 python tools/mock_async_probe.py [async 0/1] [sorts]"""
import os, sys, tempfile, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, uid, d, q, use_async, sorts):
    sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
    os.environ.update(GLU_HIP_RCCL_LIB=os.path.join(ROOT, "tests/cpp/bin/libmock_rccl.so"), GLU_MOCK_RCCL_DIR=d,
                      GLU_MOCK_RCCL_TIMEOUT_S="15", GLU_MOCK_RCCL_VERBOSE="1")
    if use_async:
        os.environ["GLU_MOCK_RCCL_ASYNC"] = "1"
    import numpy as np
    import glu_hip as G
    G.set_device(0)
    dd = G.Dist(uid, world, rank)
    n = (3 << 20) + 1000 * rank
    out = []
    busy = G.ShaderStorageBuffer(size=1 << 30)
    for i in range(sorts):
        k = np.random.default_rng(100 + rank + 10 * i).integers(0, 2**32, n, dtype=np.uint32)
        k[::9] = np.uint32(0x80000000 | rank)
        v = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(v)
        try:
            G.check(G.lib().glu_buffer_fill_u32(busy.handle(), i))  # ~0.5 ms of other work on the library queue in front of the sort
            _, _, cnt = dd.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
            G.synchronize()
            out.append(cnt)
        except Exception as e:
            out.append(str(e))
            break
    q.put((rank, out))
    q.close(); q.join_thread()
    os._exit(0)


if __name__ == "__main__":
    use_async = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    sorts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = tempfile.mkdtemp(prefix="mockprobe")
    uid = os.urandom(128)
    ps = [ctx.Process(target=worker, args=(r, 2, uid, d, q, use_async, sorts)) for r in range(2)]
    for p in ps:
        p.start()
    for _ in range(2):
        try:
            print(q.get(timeout=150), flush=True)
        except Exception as e:
            print("no result:", repr(e), flush=True)
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
