"""GPU tests of the sharded sort (BASELINE.json configs[3]).  With an RCCL process group the whole sort runs inside
libglu_hip.so (glu_dist_*: partition pass, ncclAllGather of the histograms, plan, one grouped exchange, local sort) -- here
on a 1-rank RCCL group, the only RCCL world one GPU allows; several ranks sharing the GPU go through gloo with the real
kernels doing the device work (uneven splits, plan, receive order, the receive arrays growing under a skewed plan)."""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def test_single_rank_nccl_distributed_sort(built):
    import torch
    import torch.distributed as dist
    from glu_hip import dist as D

    assert torch.cuda.is_available()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        sorter = D.DistributedRadixSort(slots=2, profile=True)
        assert sorter.native  # RCCL group: the C ABI does everything
        for n, seed in ((1 << 20, 1), (300001, 2), (4097, 3), (0, 4), (1, 5), (5 * (1 << 20) + 3, 6)):
            rng = np.random.default_rng(seed)
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
            keys[::11] = keys[0] if n else 0
            if seed == 2:
                keys &= np.uint32(0x00FFFFFF)  # one bucket holds everything
            vals = np.arange(n, dtype=np.uint32)
            kt = torch.from_numpy(keys.view(np.int32)).cuda()
            vt = torch.from_numpy(vals.view(np.int32)).cuda()
            rk, rv, cnt = sorter.sort(kt, vt)
            torch.cuda.synchronize()
            assert cnt == n
            ek, ev = O.stable_sort_pairs(keys, vals)
            assert (rk.cpu().numpy().view(np.uint32) == ek).all() and (rv.cpu().numpy().view(np.uint32) == ev).all()
            assert (kt.cpu().numpy().view(np.uint32) == keys).all()  # input untouched
        phases = sorter.phase_times()
        assert phases["sorts"] == 6 and phases["local_sort"] > 0 and phases["partition"] > 0
        # two sorts in flight on the two slots (own stream, buffers and communicator each)
        n = 3 * (1 << 20) + 11
        rng = np.random.default_rng(77)
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kt = torch.from_numpy(keys.view(np.int32)).cuda()
        vt = torch.from_numpy(vals.view(np.int32)).cuda()
        handles = [sorter.sort_async(kt, vt) for _ in range(4)]
        ek, ev = O.stable_sort_pairs(keys, vals)
        for h in handles[-2:]:
            rk, rv, cnt = h.synchronize()
            assert cnt == n and (rk.cpu().numpy().view(np.uint32) == ek).all() and (rv.cpu().numpy().view(np.uint32) == ev).all()
    finally:
        dist.destroy_process_group()


def _repartition_worker(q):
    """Own process: GLU_HIP_DIST_TEST_REPARTITION makes a one-rank glu_dist take the lower-byte fallback that a multi-rank
    sort takes when all keys share the top byte."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["GLU_HIP_DIST_TEST_REPARTITION"] = "1"
    import numpy as np
    import torch
    import glu_hip as G
    import oracle as O

    torch.cuda.set_device(0)
    G.set_device(0)
    d = G.Dist(G.dist_unique_id(), 1, 0)
    out = []
    for bits, expect_shift in ((32, 24), (24, 16), (13, 8), (5, 0), (0, 0)):
        n = 700001
        rng = np.random.default_rng(bits)
        keys = (rng.integers(0, 2**32, n, dtype=np.uint64) & ((1 << bits) - 1)).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kt = torch.from_numpy(keys.view(np.int32)).cuda()
        vt = torch.from_numpy(vals.view(np.int32)).cuda()
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            kp, vp, cnt = d.sort_ptr(kt.data_ptr(), vt.data_ptr(), n, stream.cuda_stream)
        stream.synchronize()
        import ctypes

        def read_back(ptr):  # the shard lives in the library's own arrays: wrap the pointer as a buffer and read it
            h = ctypes.c_uint32(0)
            G.check(G.lib().glu_buffer_wrap(ctypes.c_void_p(ptr), n * 4, ctypes.byref(h)))
            host = np.empty(n, dtype=np.uint32)
            G.check(G.lib().glu_buffer_read(h, host.ctypes.data_as(ctypes.c_void_p), n * 4, 0))
            G.check(G.lib().glu_buffer_destroy(h))
            return host

        gk, gv = read_back(kp), read_back(vp)
        ek, ev = O.stable_sort_pairs(keys, vals)
        ok = cnt == n and (gk == ek).all() and (gv == ev).all()
        out.append((bits, d.partition_shift(), expect_shift, bool(ok)))
    q.put(out)


def test_small_range_keys_fall_back_to_a_lower_partition_byte(built):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_repartition_worker, args=(q,))
    p.start()
    import queue

    try:
        out = q.get(timeout=120)
    except queue.Empty:
        p.join(timeout=5)
        raise AssertionError("worker produced nothing (exit code %s)" % p.exitcode)
    p.join(timeout=60)
    assert p.exitcode == 0
    for bits, shift, expect_shift, ok in out:
        assert ok, bits
        assert shift == expect_shift, (bits, shift, expect_shift)


def _gpu_worker(rank, world, port, n_local, q):
    """One of `world` processes sharing GPU 0: real device ops (HipLocalOps), gloo as the transport (RCCL refuses two
    ranks on one GPU) -- exercises uneven splits, the plan and the receive ordering with the real kernels."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import torch.distributed as dist
    from glu_hip import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(50 + rank)
        n = n_local + 1000 * rank
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        keys[::9] = np.uint32(0x80000000 | rank)  # duplicates within and across ranks
        if rank == 1:
            keys[: n // 2] |= np.uint32(0xF0000000)  # skew: rank 1 is heavy in the top buckets
        base = sum(n_local + 1000 * r for r in range(rank))
        vals = np.arange(base, base + n, dtype=np.uint32)
        sorter = D.DistributedRadixSort(slots=2)  # two slots: consecutive sorts run on alternating streams / buffers
        kt = torch.from_numpy(keys.view(np.int32).copy()).cuda()
        vt = torch.from_numpy(vals.view(np.int32).copy()).cuda()
        handles = [sorter.sort_async(kt, vt) for _ in range(3)]  # same input three times, overlapping in flight
        torch.cuda.synchronize()
        first = handles[1].synchronize()
        rk, rv, cnt = handles[2].synchronize()
        assert cnt == first[2] and bool((rk[:cnt] == first[0][:cnt]).all()) and bool((rv[:cnt] == first[1][:cnt]).all())
        q.put((rank, keys, vals, rk[:cnt].cpu().numpy().view(np.uint32).copy(), rv[:cnt].cpu().numpy().view(np.uint32).copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multi_process_one_gpu_gloo_transport(built, world):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, 1 << 20, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    all_keys = np.concatenate([r[1] for r in results])
    all_vals = np.concatenate([r[2] for r in results])
    ek, ev = O.stable_sort_pairs(all_keys, all_vals)
    gk = np.concatenate([r[3] for r in results])
    gv = np.concatenate([r[4] for r in results])
    assert gk.size == ek.size
    assert (gk == ek).all() and (gv == ev).all()


def _big_skewed_worker(rank, world, port, log2n, q):
    """2 ranks x 2^26 pairs on one GPU over gloo; 70 % of all keys fall into one top-8-bit bucket, which is never split,
    so its owner receives far more than 1.25 x its slice and the receive arrays must grow
    (DistributedRadixSort._grow_recv).  Everything is checked
    on the device against the input's generating function; only a summary travels back."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from glu_hip import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << log2n

        def key_of(idx):  # idx: int64 global indices -> int64 in [0, 2^32)
            x = (idx * 2654435761 + 12345) & 0xFFFFFFFF
            x = x ^ (x >> 15)
            x = (x * 2246822519) & 0xFFFFFFFF
            x = x ^ (x >> 13)
            x = x & 0xFFFFF0FF  # duplicates
            return torch.where(idx % 10 < 7, (x & 0x00FFFFFF) | 0xC0000000, x)  # 70 % of the keys in bucket 0xC0

        idx = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64, device="cuda")
        keys = key_of(idx)
        kt = torch.where(keys >= 2**31, keys - 2**32, keys).to(torch.int32)
        vt = torch.where(idx >= 2**31, idx - 2**32, idx).to(torch.int32)
        del idx, keys
        sorter = D.DistributedRadixSort()
        grown = []
        original = sorter._grow_recv

        def spy(slot, b, n_recv):
            grown.append(n_recv)
            return original(slot, b, n_recv)

        sorter._grow_recv = spy
        rk, rv, cnt = sorter.sort(kt, vt)
        torch.cuda.synchronize()
        k64 = rk.to(torch.int64) & 0xFFFFFFFF
        v64 = rv.to(torch.int64) & 0xFFFFFFFF
        ok_sorted = bool((k64[1:] >= k64[:-1]).all()) if cnt > 1 else True
        ok_pairs = bool((key_of(v64) == k64).all())
        eq = k64[1:] == k64[:-1]
        ok_stable = bool((v64[1:][eq] > v64[:-1][eq]).all()) if cnt > 1 else True
        first = int(k64[0]) if cnt else -1
        last = int(k64[-1]) if cnt else -1
        vsum = int(v64.sum()) if cnt else 0
        q.put((rank, cnt, ok_sorted, ok_pairs, ok_stable, first, last, vsum, len(grown)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_2p26_each_skewed_plan_grows_receive_buffers(built):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, log2n = 2, 26
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_big_skewed_worker, args=(r, world, port, log2n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    total = world << log2n
    assert sum(r[1] for r in results) == total
    assert all(r[2] and r[3] and r[4] for r in results)  # sorted, every (key, val) pair is an input pair, stable
    assert sum(r[7] for r in results) == total * (total - 1) // 2  # every index exactly once
    assert results[0][6] <= results[1][5]  # rank 0's last key <= rank 1's first key
    heavy = max(results, key=lambda r: r[1])
    assert heavy[1] > 1.25 * (1 << log2n) and heavy[8] == 1  # the skewed shard really grew the receive arrays
