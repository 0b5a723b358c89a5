"""CPU tests of the multi-GPU path (gl-radix-sort_amd/glu_hip/dist.py): the planning functions with simulated
ranks, and the full exchange with two real processes over gloo.  The per-rank device work is replaced by an
oracle-backed stand-in here (test infrastructure); on the GPU box HipLocalOps does it through libglu_hip.so."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle as O
from glu_hip import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalOps:
    """numpy/oracle stand-in for HipLocalOps (same interface), CPU tensors."""

    def partition(self, keys, vals, out_keys, out_vals, hist):
        k = keys.numpy().view(np.uint32)
        v = vals.numpy().view(np.uint32)
        b = (k >> 24).astype(np.int64)
        order = np.argsort(b, kind="stable")
        out_keys.numpy().view(np.uint32)[:k.size] = k[order]
        out_vals.numpy().view(np.uint32)[:k.size] = v[order]
        hist.numpy()[:] = np.bincount(b, minlength=256).astype(np.int32)

    def sort(self, keys, vals, count):
        k, v = O.stable_sort_pairs(keys.numpy().view(np.uint32)[:count], vals.numpy().view(np.uint32)[:count])
        keys.numpy().view(np.uint32)[:count] = k
        vals.numpy().view(np.uint32)[:count] = v


class OracleSegmentedOps(OracleLocalOps):
    """The same stand-in with the segmented local sort of the device ops (HipLocalOps.sort_segments): every segment = its
    pieces laid end to end in the order they are listed, stably sorted by the low key_bits bits, segments in order."""

    calls = 0

    def sort_segments(self, in_keys, in_vals, out_keys, out_vals, count, begin, length, seg, nseg, key_bits):
        type(self).calls += 1
        k = in_keys.numpy().view(np.uint32)
        v = in_vals.numpy().view(np.uint32)
        ok, ov = out_keys.numpy().view(np.uint32), out_vals.numpy().view(np.uint32)
        at = 0
        for g in range(nseg):
            idx = [i for i in range(len(seg)) if seg[i] == g]
            if not idx:
                continue
            gk = np.concatenate([k[int(begin[i]):int(begin[i] + length[i])] for i in idx])
            gv = np.concatenate([v[int(begin[i]):int(begin[i] + length[i])] for i in idx])
            sk, sv = O.stable_sort_pairs(gk, gv, key_bits)
            ok[at:at + sk.size] = sk
            ov[at:at + sk.size] = sv
            at += sk.size
        assert at == count
        k[:count] = 0xDEADBEEF  # the input arrays are scratch for the device ops: nobody may rely on them afterwards


def make_keys(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 2**32, n, dtype=np.uint32)
    if kind == "dups":
        return (rng.integers(0, 50, n, dtype=np.uint32) << 22) | rng.integers(0, 3, n, dtype=np.uint32)
    if kind == "hot":  # one hot bucket + sprinkles
        k = np.full(n, 0x7F000000, dtype=np.uint32) | rng.integers(0, 4, n, dtype=np.uint32)
        k[::5] = rng.integers(0, 2**32, k[::5].size, dtype=np.uint32)
        return k
    return np.zeros(n, dtype=np.uint32)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_plan_is_contiguous_monotone_and_balanced(world):
    rng = np.random.default_rng(world)
    totals = rng.integers(1000, 2000, 256)
    owner = D.plan_bucket_to_rank(totals, world)
    assert owner[0] == 0 and owner[-1] == world - 1
    assert (np.diff(owner) >= 0).all() and set(owner.tolist()) == set(range(world))
    per_rank = np.array([totals[owner == r].sum() for r in range(world)])
    assert per_rank.max() <= totals.sum() / world + totals.max()
    # uniform buckets -> bucket b belongs to rank b * R / 256
    owner = D.plan_bucket_to_rank(np.full(256, 1 << 19), world)
    assert (owner == (np.arange(256) * world) // 256).all()


@pytest.mark.parametrize("world", [1, 2, 3, 8, 16])
def test_c_abi_plan_equals_the_python_plan(world):
    """glu_dist_plan_buckets / glu_dist_plan_counts (host-only entry points of libglu_hip.so: what the native sharded
    sort plans with) against the numpy plan of this module, on random, hot-bucket, empty and one-sided histograms."""
    import glu_hip as G

    rng = np.random.default_rng(world)
    for trial in range(12):
        h = rng.integers(0, 5000, (world, 256))
        if trial % 4 == 1:
            h[:, rng.integers(0, 256)] += 3000000
        if trial % 4 == 2:
            h[:, 1:] = 0
        if trial % 4 == 3:
            h[1:, :] = 0
        if trial == 11:
            h[:] = 0
        owner = D.plan_bucket_to_rank(h.sum(axis=0), world)
        for rank in range(world):
            send, recv = D.split_counts(h, owner, rank)
            c_owner, c_send, c_recv = G.dist_plan(h, world, rank)
            assert c_owner == [int(x) for x in owner]
            assert c_send == [int(x) for x in send] and c_recv == [int(x) for x in recv]


@pytest.mark.parametrize("rounds", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_c_abi_groups_of_the_exchange_in_rounds(world, rounds):
    """glu_dist_plan_groups (host only): every rank's buckets cut into `rounds` contiguous groups -- in order, together exactly
    the rank's buckets, never splitting a bucket, and balanced to within the largest bucket of the rank."""
    import glu_hip as G

    rng = np.random.default_rng(100 * world + rounds)
    for trial in range(10):
        h = rng.integers(0, 5000, (world, 256))
        if trial % 4 == 1:
            h[:, rng.integers(0, 256)] += 3000000
        if trial % 4 == 2:
            h[:, 7:] = 0
        if trial == 9:
            h[:] = 0
        owner = D.plan_bucket_to_rank(h.sum(axis=0), world)
        cuts = G.dist_plan_groups(h, world, owner, rounds)
        tot = h.sum(axis=0)
        for q in range(world):
            mine = np.nonzero(owner == q)[0]
            c = cuts[q]
            assert len(c) == rounds + 1 and all(a <= b for a, b in zip(c, c[1:]))
            if mine.size == 0:
                assert c[0] == c[-1]
                continue
            assert c[0] == mine[0] and c[-1] == mine[-1] + 1
            sizes = [int(tot[c[j]:c[j + 1]].sum()) for j in range(rounds)]
            total, biggest = int(tot[mine].sum()), int(tot[mine].max())
            assert sum(sizes) == total
            # every boundary lies within one bucket of its target
            for j in range(1, rounds):
                assert abs(sum(sizes[:j]) - total * j // rounds) <= biggest


def test_plan_skew_never_splits_a_bucket():
    totals = np.zeros(256, dtype=np.int64)
    totals[77] = 10**6
    totals[3] = 5
    owner = D.plan_bucket_to_rank(totals, 8)
    assert (np.diff(owner) >= 0).all()
    assert len(set(owner[totals > 0].tolist())) <= 2


@pytest.mark.parametrize("kind", ["uniform", "dups", "hot", "zero"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_simulated_ranks_equal_single_device_stable_sort(kind, world):
    """Pure-function simulation of R ranks: partition -> plan -> exchange (source-rank order) -> local stable sort."""
    n_local = 5000
    shards = [make_keys(kind, n_local + 17 * r, 100 + r) for r in range(world)]
    offs = np.concatenate([[0], np.cumsum([s.size for s in shards])])
    vals = [np.arange(offs[r], offs[r + 1], dtype=np.uint32) for r in range(world)]
    parts, hists = [], []
    for r in range(world):
        b = (shards[r] >> 24).astype(np.int64)
        order = np.argsort(b, kind="stable")
        parts.append((shards[r][order], vals[r][order]))
        hists.append(np.bincount(b, minlength=256))
    all_hist = np.stack(hists)
    owner = D.plan_bucket_to_rank(all_hist.sum(0), world)
    out_k, out_v = [], []
    for me in range(world):
        seg_k, seg_v = [], []
        for src in range(world):
            send, _ = D.split_counts(all_hist, owner, src)
            so = np.concatenate([[0], np.cumsum(send)])
            seg_k.append(parts[src][0][so[me]:so[me + 1]])
            seg_v.append(parts[src][1][so[me]:so[me + 1]])
        _, recv = D.split_counts(all_hist, owner, me)
        assert [s.size for s in seg_k] == recv.tolist()
        k, v = O.stable_sort_pairs(np.concatenate(seg_k), np.concatenate(seg_v))
        out_k.append(k)
        out_v.append(v)
    ek, ev = O.stable_sort_pairs(np.concatenate(shards), np.concatenate(vals))
    assert (np.concatenate(out_k) == ek).all() and (np.concatenate(out_v) == ev).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_pieces_lists_the_shard_as_it_arrives():
    """(source, bucket) order, empty pieces dropped, begins = running sum over the source-major layout."""
    rng = np.random.default_rng(5)
    world = 4
    h = rng.integers(0, 50, (world, 256))
    h[:, 100:140] = 0  # buckets without elements
    owner = D.plan_bucket_to_rank(h.sum(axis=0), world)
    for rank in range(world):
        begin, length, seg, nseg = D.shard_pieces(h, owner, rank)
        mine = np.nonzero(owner == rank)[0]
        assert nseg == mine.size and (length > 0).all()
        _, recv = D.split_counts(h, owner, rank)
        assert int(length.sum()) == int(recv.sum())
        at, want = 0, []
        for s in range(world):
            for j, b in enumerate(mine):
                if h[s, b]:
                    want.append((at, int(h[s, b]), j))
                at += int(h[s, b])
        assert list(zip(begin.tolist(), length.tolist(), seg.tolist())) == want


def _worker(rank, world, port, kind, n_local, q, segmented=False):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys = make_keys(kind, n_local + 13 * rank, 7 + rank)
        base = sum(n_local + 13 * r for r in range(rank))
        vals = np.arange(base, base + keys.size, dtype=np.uint32)
        sorter = D.DistributedRadixSort(local_ops=OracleSegmentedOps() if segmented else OracleLocalOps())
        if segmented:
            sorter.segmented_min = 1  # every shard takes the segmented local sort (the library's threshold is 2^24 pairs)
        kt = torch.from_numpy(keys.view(np.int32).copy())
        vt = torch.from_numpy(vals.view(np.int32).copy())
        rk, rv, cnt = sorter.sort(kt, vt)
        assert sorter.last_local_sort == ("segmented" if segmented and cnt else "ordinary")
        q.put((rank, rk.numpy().view(np.uint32)[:cnt].copy(), rv.numpy().view(np.uint32)[:cnt].copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("segmented", [False, True])
@pytest.mark.parametrize("kind", ["uniform", "dups", "hot"])
def test_two_process_gloo_sort_matches_oracle(kind, segmented):
    """segmented: the local sort after the exchange is the segmented one (the shard's pieces, grouped by bucket, sorted by
    their low 24 bits) instead of the ordinary sort of all 32 bits: same result."""
    import torch.multiprocessing as mp

    world, n_local = 2, 20000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_local, q, segmented)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_keys = np.concatenate([make_keys(kind, n_local + 13 * r, 7 + r) for r in range(world)])
    all_vals = np.arange(all_keys.size, dtype=np.uint32)
    ek, ev = O.stable_sort_pairs(all_keys, all_vals)
    gk = np.concatenate([r[1] for r in results])
    gv = np.concatenate([r[2] for r in results])
    assert (gk == ek).all() and (gv == ev).all()


def _pipelined_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sorter = D.DistributedRadixSort(local_ops_factory=OracleLocalOps, slots=2)
        out = []
        for step in range(5):  # alternates between the two slots; results are copied out before a slot is reused
            keys = make_keys(["uniform", "dups", "hot"][step % 3], 9000 + 100 * step + 7 * rank, 31 * step + rank)
            vals = np.arange(keys.size, dtype=np.uint32) + np.uint32(1000000 * rank)
            h = sorter.sort_async(torch.from_numpy(keys.view(np.int32).copy()), torch.from_numpy(vals.view(np.int32).copy()))
            rk, rv, cnt = h.wait()
            out.append((keys, vals, rk.numpy().view(np.uint32)[:cnt].copy(), rv.numpy().view(np.uint32)[:cnt].copy()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_slot_pipelined_sorts_over_gloo():
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(5):
        all_keys = np.concatenate([results[r][step][0] for r in range(world)])
        all_vals = np.concatenate([results[r][step][1] for r in range(world)])
        ek, ev = O.stable_sort_pairs(all_keys, all_vals)
        gk = np.concatenate([results[r][step][2] for r in range(world)])
        gv = np.concatenate([results[r][step][3] for r in range(world)])
        assert (gk == ek).all() and (gv == ev).all(), step
