"""GPU test of the multi-GPU driver with the real device ops (HipLocalOps) on a 1-rank RCCL group: partition
pass + histogram + all_to_all_single + local sort must reproduce the single-device stable sort."""
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
        sorter = D.DistributedRadixSort()
        for n, seed in ((1 << 20, 1), (300001, 2), (4097, 3)):
            rng = np.random.default_rng(seed)
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
            keys[::11] = keys[0]
            vals = np.arange(n, dtype=np.uint32)
            kt = torch.from_numpy(keys.view(np.int32)).cuda()
            vt = torch.from_numpy(vals.view(np.int32)).cuda()
            rk, rv, cnt = sorter.sort(kt, vt)
            torch.cuda.synchronize()
            assert cnt == n
            ek, ev = O.stable_sort_pairs(keys, vals)
            assert (rk.cpu().numpy().view(np.uint32) == ek).all() and (rv.cpu().numpy().view(np.uint32) == ev).all()
            assert (kt.cpu().numpy().view(np.uint32) == keys).all()  # input untouched
    finally:
        dist.destroy_process_group()


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
