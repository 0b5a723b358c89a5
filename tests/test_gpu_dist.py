"""GPU tests of the sharded sort (BASELINE.json configs[3]).  With an RCCL process group the whole sort runs inside
libglu_hip.so (glu_dist_*: partition pass, ncclAllGather of the histograms, plan, one grouped exchange, local sort) -- here
on a 1-rank RCCL group, the only RCCL world one GPU allows; several ranks sharing the GPU go through gloo with the real
kernels doing the device work (uneven splits, plan, receive order, the receive arrays growing under a skewed plan)."""
import os
import sys

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
        assert not D.DistributedRadixSort().native  # the default is the torch transport; the C-ABI sort is opted into
        sorter = D.DistributedRadixSort(slots=2, profile=True, native=True)
        assert sorter.native  # the C ABI does everything
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


def _placement_worker(q):
    """Own process (a fresh device heap and GLU_VERBOSE read at start): glu_dist_prepare for 2^27 pairs places the sorter's
    scratch, the send-side pair and the receive-side pair by measurement; a sort into them is right; everything the searches
    allocated on the way is gone."""
    import ctypes
    import io
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["GLU_VERBOSE"] = "1"
    err_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "glu_placement_%d.err" % os.getpid())
    saved = os.dup(2)
    fd = os.open(err_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(fd, 2)
    try:
        import numpy as np
        import glu_hip as G

        G.set_device(0)
        hip = ctypes.CDLL("libamdhip64.so")
        free0, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        d = G.Dist(G.dist_unique_id(), 1, 0)
        n = 1 << 27
        keys = np.random.default_rng(5).integers(0, 2**32, n, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        G.synchronize()
        assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
        d.prepare(n, n + 4096)
        free1 = ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
        kp, vp, cnt = d.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
        G.synchronize()
        out = np.empty(cnt, dtype=np.uint32)
        h = ctypes.c_uint32(0)
        G.check(G.lib().glu_buffer_wrap(ctypes.c_void_p(kp), cnt * 4, ctypes.byref(h)))
        G.check(G.lib().glu_buffer_read(h, out.ctypes.data_as(ctypes.c_void_p), cnt * 4, 0))
        G.check(G.lib().glu_buffer_destroy(h))
        ok = cnt == n and bool((out[1:] >= out[:-1]).all()) and int(out.astype(np.uint64).sum()) == int(keys.astype(np.uint64).sum())
        d.destroy()
    finally:
        os.dup2(saved, 2)
        os.close(fd)
    log = open(err_path).read()
    os.unlink(err_path)
    q.put((ok, free0.value - free1.value, log.count("scratch placement:"), log.count("pair placement:"), log.count("candidate")))


def test_prepare_places_every_array_of_the_sharded_sort(built):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_placement_worker, args=(q,))
    p.start()
    ok, held, scratch_searches, pair_searches, candidates = q.get(timeout=600)
    p.join(timeout=60)
    assert ok
    # four searches (the sorter's scratch, the send pair, the receive pair, the landing pair), at least eight candidates each
    assert scratch_searches == 1 and pair_searches == 3 and candidates >= 4 * 8
    # what prepare holds afterwards: 4 pairs of 2^27 (+ 4096) 4-byte words, tables, the communicator's buffers (0.7 GiB) and what
    # the HIP allocator keeps of freed blocks (about 1 GiB, reused by later allocations: tools/place_probe_dist.py) -- not the
    # 32+ candidates, their spacers or the calibration arrays
    assert held < 8 * ((1 << 27) + 4096) * 4 + (3 << 30), held


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


def _gpu_worker(rank, world, port, n_local, q, segmented=False):
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
        if segmented:
            sorter.segmented_min = 1  # the segmented local sort for every shard (the default threshold is 2^24 pairs)
        kt = torch.from_numpy(keys.view(np.int32).copy()).cuda()
        vt = torch.from_numpy(vals.view(np.int32).copy()).cuda()
        handles = [sorter.sort_async(kt, vt) for _ in range(3)]  # same input three times, overlapping in flight
        torch.cuda.synchronize()
        first = handles[1].synchronize()
        rk, rv, cnt = handles[2].synchronize()
        assert cnt == first[2] and bool((rk[:cnt] == first[0][:cnt]).all()) and bool((rv[:cnt] == first[1][:cnt]).all())
        assert sorter.last_local_sort == ("segmented" if segmented else "ordinary")
        q.put((rank, keys, vals, rk[:cnt].cpu().numpy().view(np.uint32).copy(), rv[:cnt].cpu().numpy().view(np.uint32).copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("segmented", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_multi_process_one_gpu_gloo_transport(built, world, segmented):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, 1 << 20, q, segmented)) for r in range(world)]
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


# ---- the multi-rank code of the C ABI, on one GPU: processes as ranks, tests/cpp/mock_rccl.cpp as the transport ---------
# RCCL refuses two ranks on one GPU, so glu_dist_* binds a test double for the nine librccl entry points
# (GLU_HIP_RCCL_LIB) that exchanges through files.  Everything else is the product: partition kernels, the histogram
# gather, the plan, the send / receive offsets of the grouped exchange, the local sort.

def _collect(q, procs, world, timeout):
    """One result per rank from the queue; gives up at once when a rank has died without delivering (instead of waiting
    out the timeout while the surviving ranks sit in a collective)."""
    import queue
    import time

    results, deadline = {}, time.time() + timeout
    while len(results) < world:
        try:
            rank, out = q.get(timeout=2)
            results[rank] = out
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank %s (exit codes %s)" % ("died" if dead else "hung", [p.exitcode for p in procs]))
    return results


def _mock_cases(world):
    """(name, expected partition shift or None, per-rank (keys, vals)) -- vals are global indices."""
    cases = []

    def build(name, shift, key_fn, sizes):
        per_rank, base = [], 0
        for r, n in enumerate(sizes):
            keys = key_fn(r, n).astype(np.uint32)
            per_rank.append((keys, np.arange(base, base + n, dtype=np.uint32)))
            base += n
        cases.append((name, shift, per_rank))

    sizes = [150001 + 1000 * r for r in range(world)]

    def uniform(r, n):
        k = np.random.default_rng(100 + r).integers(0, 2**32, n, dtype=np.uint32)
        k[::9] = np.uint32(0x80000000 | r)  # duplicates within and across ranks
        return k

    def hot_bucket(r, n):
        k = np.random.default_rng(200 + r).integers(0, 2**32, n, dtype=np.uint32)
        hot = np.arange(n) % 10 < 7
        k[hot] = (k[hot] & np.uint32(0x00FFFFFF)) | np.uint32(0xC0000000)
        return k

    def small_range(r, n):  # every key below 2^24 on every rank: the ranks agree on the next byte down
        return np.random.default_rng(300 + r).integers(0, 2**24, n, dtype=np.uint32) & np.uint32(0xFFFF0F)

    def small_on_rank0(r, n):  # only rank 0's keys are small: no fallback
        k = np.random.default_rng(400 + r).integers(0, 2**32, n, dtype=np.uint32)
        return k & np.uint32(0xFFFFFF) if r == 0 else k

    build("uniform", 24, uniform, sizes)
    build("hot_bucket", 24, hot_bucket, sizes)
    build("all_equal", 0, lambda r, n: np.full(n, 0xDEADBEEF, dtype=np.uint32), sizes)  # one rank receives everything
    build("small_range", 16, small_range, sizes)
    build("small_on_rank0", 24, small_on_rank0, sizes)
    build("rank0_empty", 24, uniform, [0] + sizes[1:])
    build("only_last_rank_has_keys", 24, uniform, [0] * (world - 1) + [70001])
    build("two_keys", None, lambda r, n: np.array([5, 3][:n], dtype=np.uint32), [2] + [0] * (world - 1))
    build("line_kernel_sizes", 24, uniform, [3 * (1 << 20) + 17 + 4096 * r for r in range(world)])
    # equal keys that straddle source ranks: five key values over all ranks (stability = source rank, then source index)
    build("few_distinct_keys", 24,
          lambda r, n: np.random.default_rng(500 + r).choice(np.array([7, 0x40000007, 0x40000008, 0xC0FFEE00, 0xFFFFFFFF], dtype=np.uint32), n),
          sizes)
    # a duplicate-heavy low half under a uniform top byte: long runs of equal keys inside every bucket, from every source
    build("duplicates_in_every_bucket", 24,
          lambda r, n: (np.random.default_rng(600 + r).integers(0, 256, n, dtype=np.uint32) << np.uint32(24)) | np.uint32(0x00ABCD00 + r % 2),
          [400003 + 64 * r for r in range(world)])
    return cases


def _mock_rank_worker(rank, world, unique_id, mock_lib, mock_dir, q, seg_mode=None, mock_async=False, rounds=None):
    import os
    import sys

    if rounds:  # the exchange in `rounds` rounds (groups of buckets), whatever the shard size
        os.environ["GLU_HIP_DIST_ROUNDS"] = str(rounds)
        os.environ["GLU_HIP_DIST_ROUNDS_MIN"] = "1"

    if mock_async:  # the test double only enqueues, like RCCL (tests/cpp/mock_rccl.cpp): stream-order bugs become wrong data
        os.environ["GLU_MOCK_RCCL_ASYNC"] = "1"
    if seg_mode is not None:
        os.environ["GLU_HIP_DIST_SEG"] = seg_mode  # "2": segmented local sort for every shard of 2^16 pairs up; "0": never

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["GLU_HIP_RCCL_LIB"] = mock_lib
    os.environ["GLU_MOCK_RCCL_DIR"] = mock_dir
    import ctypes

    import numpy as np
    import glu_hip as G
    from test_gpu_dist import _mock_cases

    G.set_device(0)
    first = G.Dist(unique_id, world, rank)
    second = G.Dist(unique_id[::-1], world, rank)  # a second communicator: two sorts in flight below
    out = []
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):  # two caller streams (non-blocking), one per object, for the back-to-back sequence below
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        streams.append(st)

    def read_back(ptr, n):
        host = np.empty(n, dtype=np.uint32)
        if n:
            h = ctypes.c_uint32(0)
            G.check(G.lib().glu_buffer_wrap(ctypes.c_void_p(ptr), n * 4, ctypes.byref(h)))
            G.check(G.lib().glu_buffer_read(h, host.ctypes.data_as(ctypes.c_void_p), n * 4, 0))
            G.check(G.lib().glu_buffer_destroy(h))
        return host

    for name, shift, per_rank in _mock_cases(world):
        keys, vals = per_rank[rank]
        kb, vb = G.ShaderStorageBuffer(keys) if keys.size else None, G.ShaderStorageBuffer(vals) if vals.size else None
        kp, vp = (kb.device_ptr(), vb.device_ptr()) if keys.size else (None, None)
        gk_ptr, gv_ptr, cnt = first.sort_ptr(kp, vp, keys.size)
        G.synchronize()
        gk, gv = read_back(gk_ptr, cnt), read_back(gv_ptr, cnt)
        # the same sort as begin / finish on two objects at once (what bench.py's depth 2 does), into caller arrays
        n1 = first.sort_begin(kp, vp, keys.size)
        n2 = second.sort_begin(kp, vp, keys.size)
        rk1, rv1 = G.ShaderStorageBuffer(size=max(n1, 1) * 4), G.ShaderStorageBuffer(size=max(n1, 1) * 4)
        rk2, rv2 = G.ShaderStorageBuffer(size=max(n2, 1) * 4), G.ShaderStorageBuffer(size=max(n2, 1) * 4)
        first.sort_finish(rk1.device_ptr(), rv1.device_ptr(), n1)
        second.sort_finish(rk2.device_ptr(), rv2.device_ptr(), n2)
        G.synchronize()
        same = n1 == cnt and n2 == cnt
        for b, ref in ((rk1, gk), (rv1, gv), (rk2, gk), (rv2, gv)):
            same = same and bool((b.get_data(np.uint32)[:cnt] == ref).all())
        # back to back with no host synchronisation in between, each object on its own caller stream: a sort of a DIFFERENT
        # input (the slice reversed) followed at once by the sort of the real one on the same object -- its partition
        # arrays, histogram rows, landing arrays and scratch are reused while the sort before is still in flight -- and the
        # other object's sort of the real input beside them.  Only stream order keeps the three apart.
        rev_k = G.ShaderStorageBuffer(np.ascontiguousarray(keys[::-1])) if keys.size else None
        rev_v = G.ShaderStorageBuffer(np.ascontiguousarray(vals[::-1])) if keys.size else None
        rkp, rvp = (rev_k.device_ptr(), rev_v.device_ptr()) if keys.size else (None, None)
        G.synchronize()
        first.sort_ptr(rkp, rvp, keys.size, stream=streams[0].value)
        b2k, b2v, b2n = second.sort_ptr(kp, vp, keys.size, stream=streams[1].value)
        b1k, b1v, b1n = first.sort_ptr(kp, vp, keys.size, stream=streams[0].value)
        assert hip.hipDeviceSynchronize() == 0  # (G.synchronize waits for the library queue only: these ran on caller streams)
        same = same and b1n == cnt and b2n == cnt
        for ptr, ref in ((b1k, gk), (b1v, gv), (b2k, gk), (b2v, gv)):
            same = same and bool((read_back(ptr, cnt) == ref).all())
        if keys.size:  # the input is untouched
            same = same and bool((kb.get_data(np.uint32) == keys).all()) and bool((vb.get_data(np.uint32) == vals).all())
        out.append((name, first.partition_shift(), gk, gv, same, first.last_local_sort(), first.last_rounds()))
    first.destroy()
    second.destroy()
    q.put((rank, out))


@pytest.mark.parametrize("seg_mode,mock_async,rounds",
                         [("2", False, None), ("0", False, None), (None, False, None), ("2", True, None), (None, True, None),
                          ("2", True, 3), ("2", False, 4), ("0", True, 2)],
                         ids=["seg-sync", "noseg-sync", "default-sync", "seg-async", "default-async", "seg-async-3rounds", "seg-sync-4rounds",
                              "noseg-async-2rounds"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_multi_rank_sort_over_mock_transport(built, world, seg_mode, mock_async, rounds, tmp_path):
    """seg_mode "2": every shard of 2^16 pairs or more takes the segmented local sort (the exchange then lands in the
    sorter's scratch and the first pass regroups the source-major shard by bucket), "0": never, None: the library's rule
    (these shards are below its 2^24 threshold).  mock_async: the test double only ENQUEUES its collectives on the caller's
    stream and returns (GLU_MOCK_RCCL_ASYNC, like the real library), so a missing stream dependency shows as wrong data;
    the synchronous mode proves offsets and the plan only.  rounds: the exchange is posted in that many rounds, one group of
    every rank's buckets each, on the side stream, and the local sort of a group runs behind its own round (what sorts of 2^24
    pairs per rank do by themselves); every round's messages, the group-major landing layout and the per-group segmented sorts
    at their places in the shard are then what the comparison with the oracle checks."""
    import torch.multiprocessing as mp

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock_lib = os.path.join(root, "tests", "cpp", "bin", "libmock_rccl.so")
    assert os.path.exists(mock_lib), "tests/cpp/bin/libmock_rccl.so is not built (make -C tests/cpp)"
    unique_id = os.urandom(128)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mock_rank_worker, args=(r, world, unique_id, mock_lib, str(tmp_path), q, seg_mode, mock_async, rounds))
             for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world, 900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for ci, (name, shift, per_rank) in enumerate(_mock_cases(world)):
        all_keys = np.concatenate([k for k, _ in per_rank])
        all_vals = np.concatenate([v for _, v in per_rank])
        ek, ev = O.stable_sort_pairs(all_keys, all_vals)
        got = [results[r][ci] for r in range(world)]
        assert all(g[0] == name for g in got)
        gk = np.concatenate([g[2] for g in got])
        gv = np.concatenate([g[3] for g in got])
        assert gk.size == ek.size, name
        assert (gk == ek).all() and (gv == ev).all(), name  # rank outputs in rank order = the single-device stable sort
        assert all(g[4] for g in got), name                 # begin / finish on two communicators gave the same shards
        if shift is not None:
            assert all(g[1] == shift for g in got), (name, [g[1] for g in got])
        for g in got:  # which local sort ran: the segmented one exactly when forced and the shard is large enough for it
            want = "segmented" if (seg_mode == "2" and g[2].size >= (1 << 16) and g[1] == 24) else "ordinary"
            assert g[5] == want, (name, g[2].size, g[1], g[5])
            # rounds are used on the top-byte partition only (a lower byte means tiny or degenerate inputs)
            assert g[6] == (rounds if rounds and g[1] == 24 else 1), (name, g[6])
        if name == "uniform":  # the plan balances: a shard exceeds its share by less than the hottest (unsplittable) bucket
            sizes = [g[2].size for g in got]
            hottest = int(np.bincount(all_keys >> 24, minlength=256).max())
            assert max(sizes) <= ek.size / world + hottest, (sizes, hottest)


def _default_settings_worker(rank, world, unique_id, mock_lib, mock_dir, q, n):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(GLU_HIP_RCCL_LIB=mock_lib, GLU_MOCK_RCCL_DIR=mock_dir, GLU_MOCK_RCCL_ASYNC="1")
    import ctypes

    import numpy as np
    import glu_hip as G

    G.set_device(0)
    d = G.Dist(unique_id, world, rank)
    keys = np.random.default_rng(900 + rank).integers(0, 2**32, n, dtype=np.uint32)
    keys[::13] = np.uint32(0x40000000 + rank)  # duplicates inside and across the ranks
    vals = np.arange(rank * n, (rank + 1) * n, dtype=np.uint32)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    d.prepare(n, n + n // 4)
    kp, vp, cnt = d.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
    G.synchronize()
    out = []
    for ptr in (kp, vp):
        host = np.empty(cnt, dtype=np.uint32)
        h = ctypes.c_uint32(0)
        G.check(G.lib().glu_buffer_wrap(ctypes.c_void_p(ptr), cnt * 4, ctypes.byref(h)))
        G.check(G.lib().glu_buffer_read(h, host.ctypes.data_as(ctypes.c_void_p), cnt * 4, 0))
        G.check(G.lib().glu_buffer_destroy(h))
        out.append(host)
    q.put((rank, (out[0], out[1], d.last_local_sort(), d.last_rounds())))
    d.destroy()


@pytest.mark.parametrize("world", [2, 8])
def test_native_ranks_with_the_library_defaults_at_production_size(built, tmp_path, world):
    """Two and EIGHT ranks of 2^24 pairs each with NOTHING overridden: the sizes from which the library by itself sorts the shard
    with the segmented sort that ends in LDS and -- from four ranks up -- posts the exchange in rounds (three) and sorts group by
    group, after a prepare that placed every array by measurement -- over the asynchronous test double, against the oracle.
    (World 8 is the shape of BASELINE.json configs[3] at an eighth of its size: eight processes share the one GPU.)"""
    import torch.multiprocessing as mp

    n = 1 << 24
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock_lib = os.path.join(root, "tests", "cpp", "bin", "libmock_rccl.so")
    unique_id = os.urandom(128)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_default_settings_worker, args=(r, world, unique_id, mock_lib, str(tmp_path), q, n)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world, 900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    all_keys = np.concatenate([np.random.default_rng(900 + r).integers(0, 2**32, n, dtype=np.uint32) for r in range(world)])
    for r in range(world):
        all_keys[r * n:(r + 1) * n:13] = np.uint32(0x40000000 + r)
    ek, ev = O.stable_sort_pairs(all_keys, np.arange(world * n, dtype=np.uint32))
    gk = np.concatenate([results[r][0] for r in range(world)])
    gv = np.concatenate([results[r][1] for r in range(world)])
    assert gk.size == ek.size and (gk == ek).all() and (gv == ev).all()
    # three rounds from four ranks up, one below; a rank's local sort is the segmented one from 2^24 pairs up (the split is not
    # exactly even, so the ranks may well differ: one lands its rounds in the group-major layout, another in the source-major one)
    for r in range(world):
        assert results[r][3] == (3 if world >= 4 else 1), results[r][2:]
        assert results[r][2] == ("segmented" if results[r][0].size >= (1 << 24) else "ordinary"), (results[r][0].size, results[r][2])


def _mock_fault_worker(rank, world, unique_id, mock_lib, mock_dir, q, fault, mock_async=True):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(GLU_HIP_RCCL_LIB=mock_lib, GLU_MOCK_RCCL_DIR=mock_dir, GLU_MOCK_RCCL_ASYNC="1" if mock_async else "0",
                      GLU_MOCK_RCCL_TIMEOUT_S="20")
    if fault:
        os.environ["GLU_HIP_DIST_TEST_FAULT"] = fault
    import ctypes

    import numpy as np
    import glu_hip as G

    G.set_device(0)
    hip = ctypes.CDLL("libamdhip64.so")
    d = G.Dist(unique_id, world, rank)
    n = 3 * (1 << 20) + 1000 * rank
    verdicts = []
    busy = G.ShaderStorageBuffer(size=1 << 30)
    for attempt in range(3):  # a different input every time: a histogram left over from the sort before is a wrong one
        keys = np.random.default_rng(50 + 7 * attempt + rank).integers(0, 2**32 >> (8 * attempt), n, dtype=np.uint32) << np.uint32(8 * attempt)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        G.synchronize()
        try:
            # half a millisecond of other work on the library queue in front of the sort: the partition's kernels wait behind
            # it, and a side stream that is NOT ordered behind them gathers the histogram long before it exists
            G.check(G.lib().glu_buffer_fill_u32(busy.handle(), attempt))
            kp, vp, cnt = d.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
            assert hip.hipDeviceSynchronize() == 0  # (every stream: a fault may have moved work off the library queue)
            host = np.empty(cnt, dtype=np.uint32)
            h = ctypes.c_uint32(0)
            if cnt:
                G.check(G.lib().glu_buffer_wrap(ctypes.c_void_p(kp), cnt * 4, ctypes.byref(h)))
                G.check(G.lib().glu_buffer_read(h, host.ctypes.data_as(ctypes.c_void_p), cnt * 4, 0))
                G.check(G.lib().glu_buffer_destroy(h))
            verdicts.append(("sorted" if (host[1:] >= host[:-1]).all() else "unsorted", int(cnt), int(host.astype(np.uint64).sum()),
                             int(host[0]) if cnt else -1, int(host[-1]) if cnt else -1, int(keys.astype(np.uint64).sum())))
        except G.GluError as e:
            verdicts.append(("error", str(e)))
            break  # (the ranks may no longer agree on what comes next)
    q.put((rank, verdicts))
    q.close()
    q.join_thread()  # (the result is on its way before the process leaves)
    os._exit(0)  # (after a fault the object may be unusable: no orderly destroy)


@pytest.mark.parametrize("fault,mock_async,expect_clean", [(None, True, True), ("no_hist_wait", True, False),
                                                           ("local_sort_unordered", True, False), ("local_sort_unordered", False, True)],
                         ids=["no-fault-async", "no_hist_wait-async", "local_sort_unordered-async", "local_sort_unordered-sync"])
def test_async_transport_catches_a_missing_stream_dependency(built, fault, mock_async, expect_clean, tmp_path):
    """Negative controls of the test double (GLU_HIP_DIST_TEST_FAULT leaves a stream dependency out of the product on purpose).
    no_hist_wait: the histogram all-gather (side stream) is not ordered behind the partition's count + scan kernels; with
    half a millisecond of other work queued in front of the sort the gather reads the histogram of the sort BEFORE: the plan is
    made from it and elements are lost and duplicated.  local_sort_unordered: the local sort runs on the side stream instead of
    behind the exchange.  With a transport that has moved the bytes by the time the call returns (the synchronous double) that
    is still RIGHT -- the blind spot the asynchronous double was built for; with one that only enqueues (the asynchronous
    double, RCCL) the sort reads landing arrays nothing has arrived in.  Without a fault the same three sorts are right."""
    import queue

    import torch.multiprocessing as mp

    world = 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock_lib = os.path.join(root, "tests", "cpp", "bin", "libmock_rccl.so")
    unique_id = os.urandom(128)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mock_fault_worker, args=(r, world, unique_id, mock_lib, str(tmp_path), q, fault, mock_async)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world, 300)
    for p in procs:
        p.join(timeout=60)
    total = sum(3 * (1 << 20) + 1000 * r for r in range(world))
    # clean = every shard ascending, the shards in rank order ascending across their borders, and together exactly the keys
    # that went in (count and sum): a plan made from a stale histogram loses and duplicates elements
    clean = all(len(v) == 3 and all(x[0] == "sorted" for x in v) for v in results.values())
    for i in range(3):
        if not clean:
            break
        shards = [results[r][i] for r in range(world)]
        clean = sum(x[1] for x in shards) == total and sum(x[2] for x in shards) == sum(x[5] for x in shards)
        filled = [x for x in shards if x[1]]
        clean = clean and all(a[4] <= b[3] for a, b in zip(filled, filled[1:]))
    if expect_clean:
        assert clean, results
    else:
        assert not clean, "the %s transport did not notice the fault %s: %r" % ("asynchronous" if mock_async else "synchronous", fault, results)


def _mock_failure_worker(rank, world, unique_id, mock_lib, mock_dir, q, env, mock_async=False):
    import os
    import sys

    if mock_async:
        os.environ["GLU_MOCK_RCCL_ASYNC"] = "1"

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "gl-radix-sort_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["GLU_HIP_RCCL_LIB"] = mock_lib
    os.environ["GLU_MOCK_RCCL_DIR"] = mock_dir
    os.environ.update(env)
    import numpy as np
    import glu_hip as G

    G.set_device(0)
    d = G.Dist(unique_id, world, rank)
    n = 100000 + 10 * rank
    rng = np.random.default_rng(900 + rank)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    if env.get("GLU_HIP_DIST_TEST_SHARD_LIMIT"):
        keys[: n * 9 // 10] |= np.uint32(0xFF000000)  # nine tenths of every slice go to the last rank: over the lowered limit
    vals = np.arange(n, dtype=np.uint32)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    outcomes = []
    for attempt in range(2):
        try:
            d.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
            G.synchronize()
            outcomes.append(("ok", ""))
        except G.GluError as e:
            outcomes.append((e.status, str(e)))
    # the same through begin / finish with a rank that gives up in between (capacity 0)
    try:
        cnt = d.sort_begin(kb.device_ptr(), vb.device_ptr(), n)
        if rank == 1 and not env:
            d.sort_finish(None, None, 0)
        else:
            rk, rv = G.ShaderStorageBuffer(size=max(cnt, 1) * 4), G.ShaderStorageBuffer(size=max(cnt, 1) * 4)
            d.sort_finish(rk.device_ptr(), rv.device_ptr(), cnt)
        G.synchronize()
        outcomes.append(("ok", ""))
    except G.GluError as e:
        outcomes.append((e.status, str(e)))
    # and the object still sorts afterwards when nothing is wrong any more
    if not env:
        try:
            _, _, cnt = d.sort_ptr(kb.device_ptr(), vb.device_ptr(), n)
            G.synchronize()
            outcomes.append(("ok", cnt))
        except G.GluError as e:
            outcomes.append((e.status, str(e)))
    d.destroy()
    q.put((rank, outcomes))


@pytest.mark.parametrize("mock_async", [False, True], ids=["sync", "async"])
@pytest.mark.parametrize("env", [{"GLU_HIP_DIST_TEST_SHARD_LIMIT": "150000"}, {"GLU_HIP_DIST_TEST_FAIL": "begin:2"},
                                 {"GLU_HIP_DIST_TEST_FAIL": "finish:0"}, {}])
def test_native_failures_are_collective(built, env, mock_async, tmp_path):
    """One rank's shard over the (test-lowered) limit, a rank whose allocation fails before the histogram exchange or before
    the data exchange, a rank that cannot provide receive arrays: EVERY rank returns a failure from that call and none
    hangs in a collective; with nothing wrong any more the same objects sort."""
    import torch.multiprocessing as mp
    import queue

    world = 3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock_lib = os.path.join(root, "tests", "cpp", "bin", "libmock_rccl.so")
    unique_id = os.urandom(128)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mock_failure_worker, args=(r, world, unique_id, mock_lib, str(tmp_path), q, env, mock_async))
             for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(q, procs, world, 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import glu_hip as G

    for r in range(world):
        out = results[r]
        if env.get("GLU_HIP_DIST_TEST_SHARD_LIMIT"):
            assert [o[0] for o in out] == [G.GLU_ERROR_INVALID_ARGUMENT] * 3, (r, out)
            assert all("would receive" in o[1] and "no rank sorts" in o[1] for o in out), out
            named = {o[1].split("rank ")[1].split(" ")[0] for o in out}
            assert len(named) == 1, out  # every call names the same rank ...
            all_named = globals().setdefault("_named_ranks", set())
            all_named |= named
        elif env:
            assert [o[0] for o in out] == [G.GLU_ERROR_OUT_OF_MEMORY] * 3, (r, out)
            failing = int(env["GLU_HIP_DIST_TEST_FAIL"].split(":")[1])
            assert all(("injected failure" in o[1]) == (r == failing) for o in out), (r, out)
        else:
            assert out[0][0] == "ok" and out[1][0] == "ok", (r, out)
            assert out[2][0] == G.GLU_ERROR_INVALID_ARGUMENT, (r, out)  # rank 1 gave up: nobody exchanged
            assert ("receive arrays hold 0 pairs" in out[2][1]) == (r == 1), (r, out)
            assert out[3][0] == "ok", (r, out)
    if env.get("GLU_HIP_DIST_TEST_SHARD_LIMIT"):
        assert len(globals().pop("_named_ranks")) == 1  # ... on every rank
    if not env:
        assert sum(results[r][3][1] for r in range(world)) == sum(100000 + 10 * r for r in range(world))


@pytest.mark.parametrize("mock_async", ["0", "1"], ids=["sync", "async"])
def test_bench_multi_gpu_command_line_rehearsal(built, mock_async):
    """The driver's N > 1 bench launch (torchrun, one rank per process) with 2 ranks sharing the GPU: gloo process group,
    glu_dist over the file transport.  Not a measurement -- it checks that the launch, the collectives, the verification
    and the JSON line of bench.py's multi-GPU branch work."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAFT_REPO_ROOT=root, GLU_MOCK_RCCL_ASYNC=mock_async)  # (async: the depth-2 pair of communicators
    #                                                                                  really has two exchanges in flight)
    if mock_async == "1":
        env["GLU_HIP_DIST_ROUNDS_MIN"] = "1"  # ... and the one-at-a-time sorts post their exchange in rounds, as large shards do
        env["GLU_HIP_DIST_ROUNDS"] = "3"      # from four ranks up by default: asked for here
    p = subprocess.run(["bash", os.path.join(root, "tools", "rehearse_multi_gpu.sh"), "2", "20"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["native_c_abi"] is True
    # the line's value is the one-sort-at-a-time figure, like the N = 1 line; the two-in-flight throughput stands beside it
    # (round 6: ... the BETTER of the exchange in rounds and one grouped exchange, named in value_is; both figures stay on the line)
    best = max(line["value_depth1"], line.get("value_depth1_one_round", 0.0))
    assert line["pipeline_depth"] == 2 and line["value"] == best and line["value_depth2"] > 0 and "rehearsal" in line
    if "value_depth1_one_round" in line:
        assert ("one grouped" in line["value_is"]) == (line["value_depth1_one_round"] > line["value_depth1"]), line["value_is"]
    # ... and so do this run's own single-GPU figures, taken the same way, with the speed-ups against them
    if mock_async == "1":  # depth 1: three rounds (and the one-round figure beside it); depth 2: one round per sort
        assert line["exchange_rounds"] == 3 and line["value_depth1_one_round"] > 0
    else:
        assert line["exchange_rounds"] == 1 and "value_depth1_one_round" not in line
    one = line["one_gpu"]
    assert one["verified"] is True and one["value_depth1"] > 0 and one["value_depth2"] > 0
    for d in (1, 2):
        assert abs(line["speedup_vs_1gpu_depth%d" % d] - line["value_depth%d" % d] / one["value_depth%d" % d]) < 2e-3
    assert line["local_sort"] in ("segmented", "ordinary")
    assert line["phases_ms_rank0"]["sorts"] == line["steps"]
    # the torch.distributed transport was measured first (it is the fallback line) and stands beside the native figures
    assert line["torch_transport"]["verified"] is True and line["torch_transport"]["value"] > 0


def test_bench_plain_command_launches_its_own_ranks(built):
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE (the shape of the driver's N = 1 command): bench.py
    starts torchrun as a child process, forwards rank 0's one JSON line and returns the child's exit code."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GRAFT_REPO_ROOT=root, PLAIN="1")
    p = subprocess.run(["bash", os.path.join(root, "tools", "rehearse_multi_gpu.sh"), "2", "18"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "launching 2 ranks" in p.stderr
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout  # exactly one line on stdout: the JSON
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["native_c_abi"] is True and line["value"] > 0


def test_bench_plain_command_returns_the_ranks_failure(built):
    """... and a launch whose ranks fail (here: no transport library for the rehearsal) prints no line and does not return 0."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GLU_HIP_RCCL_LIB")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--log2-keys", "16",
                        "--rehearse-one-gpu"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert p.returncode != 0 and p.stdout.strip() == "", (p.returncode, p.stdout)


def test_bench_multi_gpu_falls_back_when_the_native_transport_hangs(built):
    """bench.py's safety net for the first real multi-GPU run: rank 1 never arrives in the native transport
    (GLU_BENCH_TEST_NATIVE_HANG), the watchdog fires, rank 0 prints the torch.distributed transport's line with
    `native_error`, every rank exits 0."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAFT_REPO_ROOT=root, GLU_BENCH_TEST_NATIVE_HANG="1", GLU_BENCH_NATIVE_DEADLINE_S="20")
    p = subprocess.run(["bash", os.path.join(root, "tools", "rehearse_multi_gpu.sh"), "2", "18"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["native_c_abi"] is False
    assert "did not finish" in line["native_error"] and line["value"] > 0

