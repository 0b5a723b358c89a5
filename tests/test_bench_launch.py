"""bench.py's launcher for `python bench.py --gpus N` (no torchrun, no WORLD_SIZE): on a machine without a GPU the ranks it
starts all refuse ("no CPU fallback"), so what can be checked here is the launch itself -- a child torchrun with N ranks on
127.0.0.1, nothing on stdout, the ranks' failure as the exit code.  The same command with ranks that succeed runs in
tests/test_gpu_dist.py::test_bench_plain_command_launches_its_own_ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_command_starts_ranks_and_returns_their_exit_code():
    import torch

    if torch.cuda.is_available():
        pytest.skip("on a GPU box the GPU test runs the same command to completion")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert "launching 2 ranks" in p.stderr and "--nproc-per-node 2" in p.stderr and "--master-addr 127.0.0.1" in p.stderr
    assert p.stderr.count("no CPU fallback") == 2  # both ranks started and refused
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_mismatch_between_gpus_and_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr


def test_pair_fingerprint_is_a_multiset_check_of_key_value_pairs():
    """The N > 1 leg of bench.py proves that values are still with their keys through two all-reduced sums over pair hashes
    (round 5): equal under any permutation of the pairs, different as soon as one value changes hands or a pair is lost."""
    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(3)
    k = torch.randint(-2**31, 2**31, (5000,), dtype=torch.int32, generator=g)
    k[::7] = k[0]  # duplicates
    v = torch.arange(5000, dtype=torch.int32)
    p = torch.randperm(5000, generator=g)
    a = bench.pair_fingerprint(torch, k, v)
    assert bool((bench.pair_fingerprint(torch, k[p], v[p]) == a).all())
    # the fingerprints of two halves add up to the whole's (what the all-reduce over ranks relies on)
    assert bool((bench.pair_fingerprint(torch, k[:1234], v[:1234]) + bench.pair_fingerprint(torch, k[1234:], v[1234:]) == a).all())
    v2 = v.clone()
    v2[0], v2[7] = v[7], v[0]  # two pairs of EQUAL keys trade values: the multiset of pairs stays the same ...
    assert bool((bench.pair_fingerprint(torch, k, v2) == a).all())
    v2[0], v2[1] = v[1], v[0]  # ... two pairs of different keys do not
    assert not bool((bench.pair_fingerprint(torch, k, v2) == a).all())
    assert not bool((bench.pair_fingerprint(torch, k[1:], v[1:]) == a).all())
