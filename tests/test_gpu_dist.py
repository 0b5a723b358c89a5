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
