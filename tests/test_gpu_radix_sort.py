"""GPU parity tests of the hot path (glu::RadixSort::operator(), reference glu/RadixSort.hpp:273-334), called
through the C ABI of libglu_hip.so and compared bit-for-bit with the oracle."""
import os

import numpy as np
import pytest

import oracle as O
from conftest import fnv1a64

pytestmark = pytest.mark.gpu

DIGIT_BITS = [4, 8]


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    print(built.device_info())
    return built


def gpu_sort(G, keys, vals, num_steps=0, bits=8, key_bytes=4):
    sorter = G.RadixSort(digit_bits=bits)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    sorter(kb, vb, keys.size, num_steps, key_bytes=key_bytes)
    return kb.get_data(keys.dtype), vb.get_data(np.uint32)


def test_native_library_is_loaded(G):
    G.RadixSort()
    maps = open("/proc/self/maps").read()
    assert "libglu_hip.so" in maps


def test_the_drivers_smoke_entry_passes(G):
    """__graft_entry__.smoke() is what the driver runs on the GPU box before the bench; it asserts outcomes (ended in LDS / refused)
    that a change of the plan's rules can flip (round 6: it did, and no test said so)."""
    import __graft_entry__ as entry

    entry.smoke()


@pytest.mark.parametrize("bits", DIGIT_BITS)
def test_reference_test_inputs_match_literal_oracle(G, golden, bits):
    """The reference's own test inputs (radix_sort_tests.cpp:88-158) with vals = iota: keys AND values must equal
    the literal restatement of the GLSL algorithm; checksums must equal the committed golden ones."""
    sums = {(c["n"], c["max"]): c for c in golden["checksums"]["cases"]}
    for c in golden["reference"]["radix_sort_tests"]["cases"]:
        n = c["n"]
        keys = O.minstd_sample(1, n, c["min"], c["max"])
        vals = np.arange(n, dtype=np.uint32)
        gk, gv = gpu_sort(G, keys, vals, bits=bits)
        ref = O.radix_sort_reference(keys, vals)
        assert (gk == ref["result_keys"]).all() and (gv == ref["result_vals"]).all(), n
        s = sums[(n, c["max"])]
        assert fnv1a64(gk) == s["sorted_keys_fnv1a64"] and fnv1a64(gv) == s["sorted_vals_fnv1a64"]
        # the reference's assertions
        assert (np.diff(gk.astype(np.int64)) >= 0).all() and (np.sort(keys) == gk).all()
        # the reference's dummy all-zero values
        gk0, gv0 = gpu_sort(G, keys, np.zeros(n, dtype=np.uint32), bits=bits)
        assert (gk0 == gk).all() and not gv0.any()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192,
                               8193, 100000, (1 << 20), (1 << 20) + 1, 3 * 4096 * 1024 + 77])
def test_sizes_full_32_bit_keys(G, bits, n):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = rng.integers(0, 2**32, n, dtype=np.uint32)
    if n <= 1:
        kb, vb = G.ShaderStorageBuffer(size=4), G.ShaderStorageBuffer(size=4)
        G.RadixSort(digit_bits=bits)(kb, vb, n)  # early-out, RadixSort.hpp:278
        return
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("key_bytes", [4, 8])
@pytest.mark.parametrize("distinct", [2, 3, 7, 9, 20])
def test_few_distinct_keys_in_random_order(G, bits, key_bytes, distinct):
    """A handful of key values in random order: the count kernels take their duplicate-peeling path (wave_tally in
    radix_sort_kernels.hpp: groups of >= 8 equal digits in a wave are added by one lane; 7 and 9 distinct values sit on
    both sides of that limit), in unplanned (2^21) and planned (2^22) sorts."""
    for n in ((1 << 21) + 12345, (1 << 22) + 5):
        rng = np.random.default_rng(distinct * 1000 + bits + key_bytes + (n & 1))
        dt = np.uint64 if key_bytes == 8 else np.uint32
        palette = rng.integers(0, 2 ** (8 * key_bytes), distinct, dtype=dt)
        keys = palette[rng.integers(0, distinct, n)]
        vals = np.arange(n, dtype=np.uint32)
        gk, gv = gpu_sort(G, keys, vals, bits=bits, key_bytes=key_bytes)
        ek, ev = O.stable_sort_pairs(keys, vals)
        assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("kind", ["zero", "ones", "few", "sorted", "reversed", "low_bits", "high_bits", "two_values"])
def test_key_distributions_are_stable(G, bits, kind):
    n = 300007
    rng = np.random.default_rng(3)
    keys = {
        "zero": np.zeros(n, dtype=np.uint32),
        "ones": np.full(n, 0xFFFFFFFF, dtype=np.uint32),
        "few": rng.integers(0, 10, n, dtype=np.uint32),
        "sorted": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32)),
        "reversed": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32))[::-1].copy(),
        "low_bits": rng.integers(0, 16, n, dtype=np.uint32),
        "high_bits": rng.integers(0, 16, n, dtype=np.uint32) << 28,
        "two_values": np.where(rng.integers(0, 2, n) == 0, 0x80000000, 0x7FFFFFFF).astype(np.uint32),
    }[kind]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("kind", ["zero", "few", "sorted", "high_bits", "mixed_runs"])
def test_key_distributions_large_tile_geometry(G, bits, kind):
    """Same, at a size that uses the 1024-thread / large-tile kernels (and, for 8-bit digits, the carry of partial
    64-byte blocks across tiles): runs of every length, from one element to whole tiles, must stay stable."""
    n = 5 * (1 << 20) + 12345
    rng = np.random.default_rng(17)
    if kind == "zero":
        keys = np.zeros(n, dtype=np.uint32)
    elif kind == "few":
        keys = rng.integers(0, 7, n, dtype=np.uint32) * np.uint32(0x01010101)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 2**32, n, dtype=np.uint32))
    elif kind == "high_bits":
        keys = rng.integers(0, 256, n, dtype=np.uint32) << 24
    else:
        # long stretches of one digit interleaved with uniform stretches: run lengths 1 .. tile size
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        for start in range(0, n, 200000):
            keys[start:start + 70000] = keys[start]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("kind", ["uniform", "one_per_tile", "sparse_digits", "sawtooth", "two_values", "all_equal_no_plan"])
def test_line_kernel_carry_cases(G, bits, kind, monkeypatch):
    """The 128-byte-line scatter (radix_scatter_lines.hpp) keeps up to 31 elements per digit in LDS between tiles:
    digits that get a handful of elements per tile (a carry that lives across many tiles without completing a line),
    digits that get whole tiles, runs that end exactly on line boundaries, a partial last tile, and workgroup range
    boundaries inside a line (n is not a multiple of anything)."""
    n = 256 * 10240 * 2 + 4321
    rng = np.random.default_rng(41)
    if kind == "uniform":
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    elif kind == "one_per_tile":
        # almost everything in one digit value, every other value about once per tile
        keys = np.full(n, 0x55555555, dtype=np.uint32)
        idx = rng.integers(0, n, n // 40)
        keys[idx] = rng.integers(0, 2**32, idx.size, dtype=np.uint32)
    elif kind == "sparse_digits":
        keys = (rng.integers(0, 5, n, dtype=np.uint32) * np.uint32(0x33333333)) ^ (rng.integers(0, 2, n, dtype=np.uint32) << 31)
    elif kind == "sawtooth":
        keys = (np.arange(n, dtype=np.uint64) * 32 % (1 << 32)).astype(np.uint32)  # runs of exactly one line per digit
    elif kind == "two_values":
        keys = np.where(rng.integers(0, 2, n) == 0, 0x80000000, 0x7FFFFFFF).astype(np.uint32)
    else:
        monkeypatch.setenv("GLU_HIP_SORT_NO_PLAN", "1")  # constant digits must go through the scatter, not be skipped
        keys = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("kind", ["uniform", "one_per_tile", "low_word_only", "two_values"])
def test_line_kernel_u64_carry_cases(G, bits, kind):
    """64-bit keys through the line kernel (16-element granules): the same carry situations as the 32-bit test."""
    n = 256 * 8192 * 2 + 999
    rng = np.random.default_rng(43)
    if kind == "uniform":
        keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    elif kind == "one_per_tile":
        keys = np.full(n, 0x5555555555555555, dtype=np.uint64)
        idx = rng.integers(0, n, n // 40)
        keys[idx] = rng.integers(0, 2**64, idx.size, dtype=np.uint64)
    elif kind == "low_word_only":
        keys = rng.integers(0, 2**20, n, dtype=np.uint64)  # the high passes are constant (skipped by the plan)
    else:
        keys = np.where(rng.integers(0, 2, n) == 0, 0x8000000000000000, 0x7FFFFFFFFFFFFFFF).astype(np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits, key_bytes=8)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all() and (gv == vals[order]).all()


@pytest.mark.parametrize("blocks", [1, 2, 7, 100, 255])
def test_line_kernel_with_fewer_workgroups(G, blocks, monkeypatch):
    """GLU_HIP_SORT_BLOCKS caps the grid: other range boundaries (first / last partial lines of a range), many tiles per
    workgroup, a single workgroup that owns everything."""
    monkeypatch.setenv("GLU_HIP_SORT_BLOCKS", str(blocks))
    n = 256 * 4 * 4096 + 777  # (just past what the small geometry takes in one round of workgroups: the line kernel's first size)
    rng = np.random.default_rng(blocks)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::5] &= np.uint32(0xFF00FFFF)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("n", [10240, 10241, 12288, 20480, 30001, 123457, 1000003])
def test_line_kernel_forced_on_small_inputs(G, bits, n, monkeypatch):
    """GLU_HIP_SORT_LARGE_MIN=1 sends every input of at least one tile through the line kernel: grids of 1 .. 100
    workgroups, one or two tiles per workgroup, a partial tile right behind the only full one (pairs, keys only, 64-bit)."""
    monkeypatch.setenv("GLU_HIP_SORT_LARGE_MIN", "1")
    monkeypatch.setenv("GLU_HIP_SORT_NO_SINGLE_BLOCK", "1")
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::3] = keys[1]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    kb = G.ShaderStorageBuffer(keys)
    G.RadixSort(digit_bits=bits).sort_keys(kb, n)
    assert (kb.get_data(np.uint32) == ek).all()
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    k64[::5] = k64[2]
    gk, gv = gpu_sort(G, k64, vals, bits=bits, key_bytes=8)
    order = np.argsort(k64, kind="stable")
    assert (gk == k64[order]).all() and (gv == vals[order]).all()


def test_line_kernel_needs_aligned_arrays_and_falls_back(G):
    """Whole-line stores need 16-byte aligned arrays; a sub-range that starts 4 bytes into an allocation takes the other
    kernel (same result), and GLU_HIP_SORT_NO_LINES=1 gives the same output as the default."""
    torch = pytest.importorskip("torch")
    n = 256 * 12288 * 2 + 5
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    keys = torch.randint(-2**31, 2**31, (n + 4,), dtype=torch.int32, device="cuda", generator=g)
    vals = torch.arange(n + 4, dtype=torch.int32, device="cuda")
    expect = {}
    for off in (0, 1, 3):
        k, v = keys.clone(), vals.clone()
        ks, vs = k[off:off + n], v[off:off + n]
        src = keys[off:off + n].clone()
        G.RadixSort().run_ptr(ks.data_ptr(), vs.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        flipped = ks ^ (-2**31)
        assert bool((flipped[1:] >= flipped[:-1]).all())
        idx = (vs.to(torch.int64) & 0xFFFFFFFF) - off
        assert bool((src[idx] == ks).all())
        eq = ks[1:] == ks[:-1]
        assert bool((idx[1:][eq] > idx[:-1][eq]).all())  # stable
        # untouched neighbours
        assert bool((k[:off] == keys[:off]).all()) and bool((k[off + n:] == keys[off + n:]).all())
        expect[off] = ks.clone()
    os.environ["GLU_HIP_SORT_NO_LINES"] = "1"
    try:
        k, v = keys.clone(), vals.clone()
        G.RadixSort().run_ptr(k.data_ptr(), v.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert bool((k[:n] == expect[0]).all())
    finally:
        del os.environ["GLU_HIP_SORT_NO_LINES"]


def test_prepare_ex_covers_every_entry_point(G):
    """glu_radix_sort_prepare_ex(count, key_bytes, with_vals): after it the matching run allocates nothing."""
    n = 300000
    rng = np.random.default_rng(9)
    s64 = G.RadixSort()
    s64.prepare_internal_buffers(n, key_bytes=8)
    size0 = s64.scratch_size()
    assert size0 >= n * 12
    keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    s64(kb, vb, n, key_bytes=8)
    order = np.argsort(keys, kind="stable")
    assert (kb.get_data(np.uint64) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
    assert s64.scratch_size() == size0
    sk = G.RadixSort()
    sk.prepare_internal_buffers(n, key_bytes=4, with_vals=False)
    size1 = sk.scratch_size()
    k32 = rng.integers(0, 2**32, n, dtype=np.uint32)
    kb = G.ShaderStorageBuffer(k32)
    sk.sort_keys(kb, n)
    assert (kb.get_data(np.uint32) == np.sort(k32)).all()
    assert sk.scratch_size() == size1 and size1 < n * 8


def test_pointer_entry_points_accept_empty_inputs(G):
    """An empty shard (torch gives data_ptr() == 0 for an empty tensor): count 0 with NULL arrays is not an error."""
    s = G.RadixSort()
    s.run_ptr(0, 0, 0)
    s.run_ptr(0, 0, 0, key_bytes=8)
    s.sort_keys_ptr(0, 0)
    s.sort_keys_ptr(0, 0, key_bytes=8)
    for key_type in G.RadixSort.KEY_TYPES:  # GLU_KEY_UINT32 .. GLU_KEY_FLOAT64
        s.sort_typed_ptr(0, 0, 0, key_type)
    s.sort_bit_range_ptr(0, 0, 0, 8, 24)
    s.sort_bit_range_ptr(0, 0, 0, 8, 24, key_bytes=8)
    with pytest.raises(G.GluError):
        s.run_ptr(0, 0, 5)
    with pytest.raises(G.GluError):
        s.sort_keys_ptr(0, 5)
    with pytest.raises(G.GluError):
        s.sort_typed_ptr(0, 0, 5, "int32")
    with pytest.raises(G.GluError):  # an invalid key type is an error whatever the count
        G.check(G.lib().glu_radix_sort_run_typed_ptr(s._h, None, None, 0, 17, None))
    with pytest.raises(G.GluError):
        s.sort_bit_range_ptr(0, 0, 5, 8, 24)
    with pytest.raises(G.GluError):
        s.sort_bit_range_ptr(0, 0, 0, 24, 8)  # so is a bad bit range


def test_destroy_releases_every_scratch_array(G):
    """64 sort objects prepared for 2^26 pairs (which includes the tables of the paired passes, 33 MiB) come and go: the
    device's free memory returns to where it was (RAII of the reference: RadixSort.hpp:194-200, gl_utils.hpp:184-188)."""
    import torch

    torch.cuda.synchronize()
    G.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    low = free0
    for i in range(64):
        s = G.RadixSort()
        s.prepare_internal_buffers(1 << 26)
        assert s.scratch_size() > 2 * 4 * (1 << 26) + (32 << 20)  # key + value scratch + the two-digit tables
        low = min(low, torch.cuda.mem_get_info()[0])
        s.destroy()
    G.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert low < free0 - (500 << 20)
    assert abs(free1 - free0) <= (8 << 20), (free0, free1)  # (the allocator's own granularity; the leak was 33 MiB per object)


def test_destroy_waits_for_work_on_a_caller_stream(G):
    """A sort enqueued on the caller's stream and the object destroyed right away: the scratch outlives the kernels."""
    import torch

    n = 6 * (1 << 20) + 5
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = O.stable_sort_pairs(keys, vals)
    side = torch.cuda.Stream()
    for _ in range(3):
        kt = torch.from_numpy(keys.view(np.int32)).cuda()
        vt = torch.from_numpy(vals.view(np.int32)).cuda()
        torch.cuda.synchronize()
        s = G.RadixSort()
        s.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, side.cuda_stream)
        s.destroy()  # frees the scratch arrays: must not happen under the running passes
        other = G.RadixSort()
        other.prepare_internal_buffers(n)  # likely to get the freed addresses
        side.synchronize()
        assert (kt.cpu().numpy().view(np.uint32) == ek).all() and (vt.cpu().numpy().view(np.uint32) == ev).all()


def test_read_plan_of_an_unplanned_sort_is_all_zero(G):
    """Below 2^22 elements a sort has no device-side plan: read_plan reports every pass as run and alone (it used to
    return whatever the plan buffer held)."""
    big = 5 * (1 << 20)
    rng = np.random.default_rng(5)
    s = G.RadixSort()
    k = rng.integers(0, 1 << 16, big, dtype=np.uint32)  # two constant digits: a planned sort that skips passes
    kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(np.arange(big, dtype=np.uint32))
    s(kb, vb, big)
    G.synchronize()
    assert any(s.read_plan(4))
    k2 = rng.integers(0, 2**32, 100000, dtype=np.uint32)
    kb2, vb2 = G.ShaderStorageBuffer(k2), G.ShaderStorageBuffer(np.arange(100000, dtype=np.uint32))
    s(kb2, vb2, 100000)
    G.synchronize()
    skipped, alone, roles = s.read_plan(4, roles=True)
    assert not any(skipped) and not any(alone) and not any(roles)
    fresh = G.RadixSort()
    fresh(kb2, vb2, 100000)
    G.synchronize()
    assert fresh.read_plan(4, roles=True) == ([0] * 4, [0] * 4, [0] * 4)


def test_small_geometry_forced_matches(G, monkeypatch):
    """GLU_HIP_SORT_SMALL=1 forces the 256-thread kernels at any size; both geometries give identical output."""
    n = 4 * (1 << 20) + 99
    rng = np.random.default_rng(23)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = O.stable_sort_pairs(keys, vals)
    monkeypatch.setenv("GLU_HIP_SORT_SMALL", "1")
    for bits in DIGIT_BITS:
        gk, gv = gpu_sort(G, keys, vals, bits=bits)
        assert (gk == ek).all() and (gv == ev).all()
    monkeypatch.delenv("GLU_HIP_SORT_SMALL")
    for bits in DIGIT_BITS:
        gk, gv = gpu_sort(G, keys, vals, bits=bits)
        assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("steps", [1, 2, 3, 4, 5, 6, 7, 8, 9, 1000])
def test_num_steps(G, bits, steps):
    """Low 4*num_steps bits only (RadixSort.hpp:289,303,331-332).  The sorted pairs equal what the reference
    produces (wherever it leaves them); they are always returned in the caller's buffers (documented deviation)."""
    n = 30011
    keys = O.minstd_sample(1, n, 0, 0xFFFFFFFF) ^ np.uint32(0x80000000)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, num_steps=steps, bits=bits)
    ref = O.radix_sort_reference(keys, vals, num_steps=steps)
    assert (gk == ref["result_keys"]).all() and (gv == ref["result_vals"]).all()


@pytest.mark.parametrize("n", [2, 777, 1024, 1025, 4096, 4097, 12288, 12289, 16383, 16384, 16385])
@pytest.mark.parametrize("steps", [0, 1, 3, 5, 8])
def test_single_workgroup_path(G, n, steps, monkeypatch):
    """n <= 16384 pairs (8192 for 64-bit keys) are sorted by one workgroup in one launch; the multi-kernel path
    (GLU_HIP_SORT_NO_SINGLE_BLOCK=1) must give the same pairs."""
    rng = np.random.default_rng(n * 10 + steps)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::5] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    ref = O.radix_sort_reference(keys, vals, num_steps=steps)
    gk, gv = gpu_sort(G, keys, vals, num_steps=steps)
    assert (gk == ref["result_keys"]).all() and (gv == ref["result_vals"]).all()
    monkeypatch.setenv("GLU_HIP_SORT_NO_SINGLE_BLOCK", "1")
    gk2, gv2 = gpu_sort(G, keys, vals, num_steps=steps)
    assert (gk2 == gk).all() and (gv2 == gv).all()
    k64 = rng.integers(0, 2**64, min(n, 8192), dtype=np.uint64)
    v64 = np.arange(k64.size, dtype=np.uint32)
    monkeypatch.delenv("GLU_HIP_SORT_NO_SINGLE_BLOCK")
    gk, gv = gpu_sort(G, k64, v64, num_steps=2 * steps, key_bytes=8)
    ek, ev = O.stable_sort_pairs(k64, v64, key_bits=64 if steps in (0, 8) else 8 * steps)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("seed", range(8))
def test_randomized_stress(G, seed):
    """Random sizes around tile / workgroup-range boundaries, random key shapes, both digit widths, random num_steps:
    every result must equal the stable sort by the masked key (= what the reference computes)."""
    rng = np.random.default_rng(1000 + seed)
    sorters = {4: G.RadixSort(digit_bits=4), 8: G.RadixSort(digit_bits=8)}
    for _ in range(12):
        base = int(rng.choice([1, 64, 4096, 12288, 16384, 12288 * 256, 4096 * 768, 1 << 20, 3 * (1 << 20)]))
        n = max(2, base * int(rng.integers(1, 3)) + int(rng.integers(-70, 70)))
        n = min(n, 7 * (1 << 20))
        shape = rng.integers(0, 6)
        if shape == 0:
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        elif shape == 1:  # few distinct values anywhere in the 32 bits
            pool = rng.integers(0, 2**32, int(rng.integers(1, 40)), dtype=np.uint32)
            keys = pool[rng.integers(0, pool.size, n)]
        elif shape == 2:  # one byte varies
            keys = (rng.integers(0, 256, n, dtype=np.uint32) << np.uint32(8 * int(rng.integers(0, 4)))) | np.uint32(0x01020304)
        elif shape == 3:  # sorted runs of random length
            keys = np.sort(rng.integers(0, 2**32, n, dtype=np.uint32))
            cut = int(rng.integers(1, n))
            keys = np.concatenate([keys[cut:], keys[:cut]])
        elif shape == 4:  # long constant stretches between random ones
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 200000)))
                keys[a:b] = keys[a]
        else:  # digits skewed towards the extremes
            keys = np.where(rng.integers(0, 10, n) < 8, 0xFFFFFFFF, rng.integers(0, 2**32, n)).astype(np.uint32)
        vals = rng.integers(0, 2**32, n, dtype=np.uint32) if rng.integers(0, 2) else np.arange(n, dtype=np.uint32)
        steps = int(rng.choice([0, 0, 0, 1, 2, 3, 5, 7, 8]))
        bits = int(rng.choice([4, 8]))
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        sorters[bits](kb, vb, n, steps)
        ek, ev = O.stable_sort_pairs(keys, vals, key_bits=32 if steps in (0, 8) else 4 * steps)
        gk, gv = kb.get_data(np.uint32), vb.get_data(np.uint32)
        assert (gk == ek).all() and (gv == ev).all(), (n, int(shape), steps, bits)


def test_prepare_then_no_growth_and_reuse(G):
    sorter = G.RadixSort()
    sorter.prepare_internal_buffers(1 << 20)
    size0 = sorter.scratch_size()
    assert size0 >= 2 * 4 * (1 << 20)
    rng = np.random.default_rng(0)
    for n in (1 << 20, 1000, 77777):
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        sorter(kb, vb, n)
        ek, ev = O.stable_sort_pairs(keys, vals)
        assert (kb.get_data(np.uint32) == ek).all() and (vb.get_data(np.uint32) == ev).all()
        assert sorter.scratch_size() == size0  # grow-only, RadixSort.hpp:237-271


def test_argument_checks(G):
    sorter = G.RadixSort()
    kb = G.ShaderStorageBuffer(np.arange(10, dtype=np.uint32))
    with pytest.raises(G.GluError) as e:
        sorter(0, kb, 10)
    assert "Invalid key buffer" in e.value.message  # RadixSort.hpp:275
    with pytest.raises(G.GluError) as e:
        sorter(kb, 0, 10)
    assert "Invalid value buffer" in e.value.message  # RadixSort.hpp:276
    with pytest.raises(G.GluError):
        sorter(kb, kb, 11)  # larger than the buffers
    with pytest.raises(G.GluError):
        G.RadixSort(digit_bits=5)


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("n", [2, 4097, 250000, 3000001])
def test_u64_keys(G, bits, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    keys[::3] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits, key_bytes=8)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    gk, gv = gpu_sort(G, keys, vals, num_steps=5, bits=bits, key_bytes=8)
    ek, ev = O.stable_sort_pairs(keys, vals, key_bits=20)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("n", [2, 1000, 12289, 1 << 20, 5 * (1 << 20) + 3])
def test_keys_only_sort(G, bits, n):
    """Keys-only entry points (the reference needs a dummy value buffer, README.md:88-89): the keys must come out exactly
    as the key half of the pair sort, for every num_steps."""
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    for steps in (0, 3, 6):
        sorter = G.RadixSort(digit_bits=bits)
        kb = G.ShaderStorageBuffer(keys)
        sorter.sort_keys(kb, n, steps)
        ek, _ = O.stable_sort_pairs(keys, np.zeros(n, dtype=np.uint32), key_bits=32 if steps == 0 else 4 * steps)
        assert (kb.get_data(np.uint32) == ek).all()
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    kb = G.ShaderStorageBuffer(k64)
    sorter = G.RadixSort(digit_bits=bits)
    sorter.sort_keys_ptr(kb.device_ptr(), n, 0, None, key_bytes=8)
    assert (kb.get_data(np.uint64) == np.sort(k64)).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("dtype", ["int32", "float32", "int64", "float64", "uint32", "uint64"])
@pytest.mark.parametrize("n", [3, 5000, 300001, 4 * (1 << 20) + 9])
def test_typed_keys(G, bits, dtype, n):
    """Signed and floating-point keys (not in the reference): natural order, stable, values travel with their keys;
    floats as a total order (-0 < +0, -inf first, +inf last)."""
    rng = np.random.default_rng(n)
    dt = np.dtype(dtype)
    if dt.kind == "f":
        keys = (rng.standard_normal(n) * 1e3).astype(dt)
        keys[rng.integers(0, n, max(1, n // 50))] = 0.0
        keys[rng.integers(0, n, max(1, n // 50))] = -0.0
        keys[rng.integers(0, n, 2)] = np.inf
        keys[rng.integers(0, n, 2)] = -np.inf
        keys[rng.integers(0, n, max(1, n // 20))] = dt.type(1.5)  # duplicates
    elif dt.kind == "i":
        info = np.iinfo(dt)
        keys = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
        keys[::7] = -3
    else:
        keys = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    vals = np.arange(n, dtype=np.uint32)
    sorter = G.RadixSort(digit_bits=bits)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    sorter.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, dtype)
    gk, gv = kb.get_data(dt), vb.get_data(np.uint32)
    # expected: stable sort by the order-preserving unsigned image of the key
    u = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
    sign = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
    if dt.kind == "i":
        image = u ^ sign
    elif dt.kind == "f":
        image = np.where(u & sign, ~u, u ^ sign)
    else:
        image = u
    order = np.argsort(image, kind="stable")
    assert (gk.view(u.dtype) == u[order]).all() and (gv == vals[order]).all()
    if dt.kind == "f":
        finite = gk[np.isfinite(gk)]
        assert (np.diff(finite) >= 0).all()
    else:
        assert (np.diff(gk.astype(object) if dt.itemsize == 8 and dt.kind == "u" else gk.astype(np.float64)) >= 0).all()
    kb2 = G.ShaderStorageBuffer(keys)
    sorter.sort_typed_ptr(kb2.device_ptr(), None, n, dtype)  # keys only
    assert (kb2.get_data(dt).view(u.dtype) == u[order]).all()


def test_raw_pointer_entry_on_torch_memory(G):
    import torch

    n = 123457
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    kt = torch.from_numpy(keys.view(np.int32)).cuda()
    vt = torch.from_numpy(vals.view(np.int32)).cuda()
    sorter = G.RadixSort()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # a caller-owned stream (handle != 0; 0 would mean "the library queue")
        assert side.cuda_stream != 0
        sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, side.cuda_stream)
        out_k, out_v = kt.cpu(), vt.cpu()  # ordered on the same stream: no device-wide sync needed
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (out_k.numpy().view(np.uint32) == ek).all() and (out_v.numpy().view(np.uint32) == ev).all()


@pytest.mark.parametrize("n", [200003, 4 * (1 << 20) + 5])
@pytest.mark.parametrize("shift,bits", [(24, 8), (28, 4), (0, 8), (13, 5), (31, 1)])
def test_partition_pass_and_histogram(G, shift, bits, n):
    import torch

    rng = np.random.default_rng(shift)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    kt = torch.from_numpy(keys.view(np.int32)).cuda()
    vt = torch.from_numpy(vals.view(np.int32)).cuda()
    ok, ov = torch.empty_like(kt), torch.empty_like(vt)
    hist = torch.zeros(1 << bits, dtype=torch.int32, device="cuda")
    sorter = G.RadixSort()
    torch.cuda.synchronize()
    sorter.partition_ptr(kt.data_ptr(), vt.data_ptr(), ok.data_ptr(), ov.data_ptr(), n, shift, bits, hist.data_ptr(), None)
    G.synchronize()  # stream None = the library queue
    d = ((keys >> shift) & ((1 << bits) - 1)).astype(np.int64)
    order = np.argsort(d, kind="stable")
    assert (ok.cpu().numpy().view(np.uint32) == keys[order]).all()
    assert (ov.cpu().numpy().view(np.uint32) == vals[order]).all()
    assert (hist.cpu().numpy() == np.bincount(d, minlength=1 << bits)).all()


def _check_sorted_properties(keys, gk, gv):
    """Size-independent properties that pin the stable sort exactly when vals = iota:
    keys ascending; gk[i] == keys[gv[i]]; gv is a permutation; gv ascending inside runs of equal keys."""
    assert (gk[1:] >= gk[:-1]).all()
    assert (keys[gv] == gk).all()
    seen = np.zeros(keys.size, dtype=bool)
    seen[gv] = True
    assert seen.all()
    eq = gk[1:] == gk[:-1]
    assert (gv[1:][eq] > gv[:-1][eq]).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
def test_full_size_2_28_properties(G, bits):
    """BASELINE.json config 3: N = 2^28 uint32 key + val, uniform random full-range keys."""
    n = 1 << 28
    rng = np.random.default_rng(0x5EED)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    _check_sorted_properties(keys, gk, gv)


def test_scratch_placement_by_measurement(G, monkeypatch):
    """prepare() of 512 MiB of keys or more tries several placements of the value scratch and keeps the fastest
    (glu_radix_sort_scratch_placement): same results as without, one measurement per growth, every temporary freed."""
    import torch

    n = 1 << 27
    rng = np.random.default_rng(123)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    monkeypatch.setenv("GLU_HIP_SCRATCH_TUNE_LIST", "768:5")
    tuned = G.RadixSort()
    tuned.prepare_internal_buffers(n)
    p = tuned.scratch_placement()
    assert p["candidates"] == 5 and 0 < p["chosen_ms"] <= p["slowest_ms"] < 1e3
    size = tuned.scratch_size()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 <= size + (64 << 20)  # the candidates, their spacers and the calibration arrays are gone
    tuned.prepare_internal_buffers(n)  # no growth: no second measurement, same arrays
    assert tuned.scratch_size() == size and tuned.scratch_placement() == p
    monkeypatch.setenv("GLU_HIP_SCRATCH_TUNE", "0")
    plain = G.RadixSort()
    plain.prepare_internal_buffers(n)
    assert plain.scratch_placement()["candidates"] == 0 and plain.scratch_size() == size
    out = []
    for s in (tuned, plain):
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        s(kb, vb, n)
        out.append((kb.get_data(np.uint32), vb.get_data(np.uint32)))
    assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()
    _check_sorted_properties(keys, out[0][0], out[0][1])
    # grow-only: a sorter prepared for a keys-only sort twice the size keeps that key scratch when pairs are prepared after it
    monkeypatch.delenv("GLU_HIP_SCRATCH_TUNE")
    grown = G.RadixSort()
    grown.prepare_internal_buffers(2 * n, with_vals=False)
    keys_only = grown.scratch_size()
    grown.prepare_internal_buffers(n)
    assert grown.scratch_placement()["candidates"] == 5 and grown.scratch_size() >= keys_only + 4 * n


def test_a_sort_never_searches_for_a_scratch_placement(G, monkeypatch):
    """Only the explicit prepare calls place the scratch by measurement.  The first sort of 2^27 pairs on an object that was
    never prepared takes two plain allocations: zero candidates, no calibration sorts (host time of the enqueue well under
    what the search costs), and a sort that outgrows what an explicit prepare had placed resets the report."""
    import time

    import torch

    n = 1 << 27
    monkeypatch.setenv("GLU_HIP_SCRATCH_TUNE_LIST", "512:4")  # (a fixed list: the search, where it runs, tries all four)
    kt = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda")
    vt = torch.arange(n, dtype=torch.int32, device="cuda")
    G.RadixSort()(G.ShaderStorageBuffer(np.arange(4096, dtype=np.uint32)), G.ShaderStorageBuffer(np.arange(4096, dtype=np.uint32)), 4096)
    torch.cuda.synchronize()
    lazy = G.RadixSort()
    stream = torch.cuda.Stream()
    t0 = time.perf_counter()
    lazy.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, stream.cuda_stream)
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    assert lazy.scratch_placement()["candidates"] == 0
    assert host_ms < 50.0, host_ms  # (the search: >= 4 x 3 calibration sorts of 2 ms + allocations)
    k = kt.cpu().numpy().view(np.uint32)
    assert (k[1:] >= k[:-1]).all()
    # an explicit prepare of a larger size does search; a later sort that outgrows it again does not, and says so
    placed = G.RadixSort()
    placed.prepare_internal_buffers(n)
    assert placed.scratch_placement()["candidates"] == 4
    n2 = n + (1 << 20)
    kt2 = torch.randint(-2**31, 2**31, (n2,), dtype=torch.int32, device="cuda")
    vt2 = torch.arange(n2, dtype=torch.int32, device="cuda")
    placed.run_ptr(kt2.data_ptr(), vt2.data_ptr(), n2, 0, stream.cuda_stream)
    torch.cuda.synchronize()
    assert placed.scratch_placement()["candidates"] == 0
    k = kt2.cpu().numpy().view(np.uint32)
    assert (k[1:] >= k[:-1]).all()


def test_full_size_2_28_duplicate_heavy(G):
    n = 1 << 28
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 1000, n, dtype=np.uint32) * np.uint32(4294967)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals)
    _check_sorted_properties(keys, gk, gv)


@pytest.mark.parametrize("bits", [8, 4])
def test_full_size_2_28_three_key_values_in_random_order(G, bits):
    """Full size with three distinct keys in random order: the pair-count kernels' duplicate-peeling loop (wave_tally), the
    16-bit two-digit counters next to their overflow, followers sent back to counting, constant digits skipped."""
    n = 1 << 28
    rng = np.random.default_rng(12 + bits)
    keys = np.array([0x00000000, 0x7F00FF01, 0xFFFFFFFF], dtype=np.uint32)[rng.integers(0, 3, n)]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, bits=bits)
    _check_sorted_properties(keys, gk, gv)


@pytest.mark.parametrize("bits", [8, 4])
def test_full_size_2_28_u64(G, bits):
    """BASELINE.json config 5: N = 2^28 uint64 keys + uint32 payload; 8-bit digits (8 passes) and the reference's
    4-bit digits (16 passes)."""
    n = 1 << 28
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, key_bytes=8, bits=bits)
    _check_sorted_properties(keys, gk, gv)


def _device_sorted_properties(torch, orig, gk, gv, chunk=1 << 28):
    """The properties of _check_sorted_properties evaluated on the device in chunks (arrays too large for the host):
    int32 tensors holding uint32 bit patterns; unsigned order = signed order after flipping bit 31."""
    n = gk.numel()
    flip = -(1 << 31)
    seen = torch.zeros(n, dtype=torch.bool, device=gk.device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        k = gk[lo:hi] ^ flip
        nxt = gk[lo + 1:min(n, hi + 1)] ^ flip
        assert bool((nxt >= k[:nxt.numel()]).all()), "keys not ascending in [%d, %d)" % (lo, hi)
        idx = gv[lo:hi].to(torch.int64) & 0xFFFFFFFF
        assert bool((orig[idx] == gk[lo:hi]).all()), "gk[i] != keys[gv[i]] in [%d, %d)" % (lo, hi)
        seen[idx] = True
        v = gv[lo:hi] ^ flip
        vn = gv[lo + 1:min(n, hi + 1)] ^ flip
        eq = nxt == k[:nxt.numel()]
        assert bool(((vn > v[:vn.numel()]) | ~eq).all()), "not stable in [%d, %d)" % (lo, hi)
        del k, nxt, idx, v, vn, eq
    assert bool(seen.all()), "vals are not a permutation"


@pytest.mark.parametrize("n,bits", [(0xFFFF0000, 8), ((1 << 31) + 12345, 4)])
def test_maximum_count_on_device(G, n, bits):
    """The largest count the ABI accepts (2^32 - 65536: 16 GiB of keys, 16 GiB of values, as much scratch) and one just
    past 2^31 (sign / 32-bit byte-offset overflows), checked by the size-independent properties on the device."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs 150 GiB of free HBM")
    gen = torch.Generator(device="cuda").manual_seed(n & 0xFFFF)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    step = 1 << 28
    for lo in range(0, n, step):  # full-range 32-bit patterns; duplicates guaranteed (n ~ 2^32 draws)
        m = min(step, n - lo)
        keys[lo:lo + m] = torch.randint(-(1 << 31), 1 << 31, (m,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    vals = torch.arange(n, dtype=torch.int64, device="cuda").to(torch.int32)  # iota as uint32 bit patterns
    orig = keys.clone()
    sorter = G.RadixSort(digit_bits=bits)
    sorter.prepare_internal_buffers(n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sorter.run_ptr(keys.data_ptr(), vals.data_ptr(), n, 0, side.cuda_stream)
    side.synchronize()
    _device_sorted_properties(torch, orig, keys, vals)
    del sorter, keys, vals, orig
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,bits", [(300001, 8), ((1 << 23) + 77, 8), ((1 << 23) + 77, 4)])
def test_sort_is_graph_capturable_and_replayable(G, n, bits):
    """run_ptr only enqueues kernels (and, for odd pass counts, one device copy) on the caller's stream once the scratch is
    prepared: it can be captured into a HIP graph and replayed on new data in the same buffers -- also a planned sort
    whose passes run in pairs (which pass counts for itself is decided on the device in every replay)."""
    import torch

    os.environ["GLU_HIP_SORT_PAIR_MIN"] = "1"
    try:
        sorter = G.RadixSort(digit_bits=bits)
    finally:
        os.environ.pop("GLU_HIP_SORT_PAIR_MIN", None)
    sorter.prepare_internal_buffers(n)
    kt = torch.empty(n, dtype=torch.int32, device="cuda")
    vt = torch.empty(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    rng = np.random.default_rng(77)
    with torch.cuda.stream(side):
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        kt.copy_(torch.from_numpy(keys.view(np.int32)))
        vt.copy_(torch.arange(n, dtype=torch.int32))
        sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, side.cuda_stream)  # warm-up outside the capture
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            keys = rng.integers(0, 2**32 if rep else 50, n, dtype=np.uint32)
            vals = np.arange(n, dtype=np.uint32)
            kt.copy_(torch.from_numpy(keys.view(np.int32)))
            vt.copy_(torch.from_numpy(vals.view(np.int32)))
            graph.replay()
            side.synchronize()
            ek, ev = O.stable_sort_pairs(keys, vals)
            assert (kt.cpu().numpy().view(np.uint32) == ek).all() and (vt.cpu().numpy().view(np.uint32) == ev).all()


def test_two_sorters_on_two_streams_concurrently(G):
    """Distinct RadixSort instances on distinct streams are independent (the reference's objects are bound to one GL
    context; here each owns its scratch): interleaved launches must not disturb each other."""
    import torch

    sizes = (1 << 22, (1 << 21) + 12345)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    sorters = [G.RadixSort(), G.RadixSort(digit_bits=4)]
    rng = np.random.default_rng(5)
    host = [(rng.integers(0, 2**32, n, dtype=np.uint32), np.arange(n, dtype=np.uint32)) for n in sizes]
    dev = [(torch.from_numpy(k.view(np.int32)).cuda(), torch.from_numpy(v.view(np.int32)).cuda()) for k, v in host]
    pristine = [(k.clone(), v.clone()) for k, v in dev]
    for s, n in zip(sorters, sizes):
        s.prepare_internal_buffers(n)
    torch.cuda.synchronize()
    for rep in range(4):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                dev[i][0].copy_(pristine[i][0])
                dev[i][1].copy_(pristine[i][1])
                sorters[i].run_ptr(dev[i][0].data_ptr(), dev[i][1].data_ptr(), sizes[i], 0, streams[i].cuda_stream)
    torch.cuda.synchronize()
    for i in (0, 1):
        ek, ev = O.stable_sort_pairs(*host[i])
        assert (dev[i][0].cpu().numpy().view(np.uint32) == ek).all() and (dev[i][1].cpu().numpy().view(np.uint32) == ev).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("key_bytes,begin,end", [(4, 0, 32), (4, 0, 24), (4, 8, 20), (4, 24, 32), (4, 5, 6), (4, 13, 13),
                                                  (8, 0, 64), (8, 28, 36), (8, 30, 51), (8, 40, 64), (8, 0, 20)])
@pytest.mark.parametrize("n", [777, 20001, 600011])
def test_bit_range_sort(G, bits, key_bytes, begin, end, n):
    """glu_radix_sort_run_bit_range_ptr: stable sort by the key bits [begin, end) only, pairs and keys-only; ranges that
    cross the two words of a 64-bit key and ranges narrower than a digit included."""
    rng = np.random.default_rng(n + begin * 64 + end)
    dt = np.uint32 if key_bytes == 4 else np.uint64
    keys = rng.integers(0, 2 ** (8 * key_bytes), n, dtype=dt)
    vals = np.arange(n, dtype=np.uint32)
    field = (keys >> dt(begin)) & dt((1 << (end - begin)) - 1) if end > begin else np.zeros(n, dtype=dt)
    order = np.argsort(field, kind="stable")
    sorter = G.RadixSort(digit_bits=bits)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    sorter.sort_bit_range_ptr(kb.device_ptr(), vb.device_ptr(), n, begin, end, None, key_bytes)
    assert (kb.get_data(dt) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
    kb2 = G.ShaderStorageBuffer(keys)
    sorter.sort_bit_range_ptr(kb2.device_ptr(), None, n, begin, end, None, key_bytes)
    assert (kb2.get_data(dt) == keys[order]).all()


def test_bit_range_argument_checks(G):
    sorter = G.RadixSort()
    kb = G.ShaderStorageBuffer(np.arange(64, dtype=np.uint32))
    for begin, end, kbytes in ((9, 8, 4), (0, 33, 4), (0, 65, 8)):
        with pytest.raises(G.GluError):
            sorter.sort_bit_range_ptr(kb.device_ptr(), None, 64, begin, end, None, kbytes)


@pytest.mark.parametrize("kind", ["constant_byte_1", "constant_bytes_0_and_2", "16_bit_keys", "all_equal"])
def test_pass_plan_on_the_small_geometry(G, kind):
    """2^22 pairs: the one size that is sorted with a device-side pass plan (passes on constant digits are skipped, the arrays'
    roles follow on the device, an odd number of executed passes is brought home) by the small geometry's kernels."""
    n = 1 << 22
    rng = np.random.default_rng(len(kind))
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    if kind == "constant_byte_1":
        keys = (keys & np.uint32(0xFFFF00FF)) | np.uint32(0x00005A00)  # three passes run: the result comes home from the scratch
    elif kind == "constant_bytes_0_and_2":
        keys = (keys & np.uint32(0xFF00FF00)) | np.uint32(0x00C30011)
    elif kind == "16_bit_keys":
        keys &= np.uint32(0xFFFF)
    else:
        keys[:] = 0xDEADBEEF
    vals = np.arange(n, dtype=np.uint32)
    s = G.RadixSort()
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    s(kb, vb, n)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (kb.get_data(np.uint32) == ek).all() and (vb.get_data(np.uint32) == ev).all()
    skipped = s.read_plan(4)[0]
    expected = {"constant_byte_1": [0, 1, 0, 0], "constant_bytes_0_and_2": [1, 0, 1, 0], "16_bit_keys": [0, 0, 1, 1], "all_equal": [1, 1, 1, 1]}[kind]
    assert [1 if x else 0 for x in skipped] == expected


@pytest.mark.parametrize("mode,threshold", [("pairs", 256 * 10240 * 3 // 2), ("keys", 256 * 16384 * 3 // 2), ("u64", 256 * 8192 * 3 // 2),
                                            ("pairs", 256 * 12288 * 3 // 2), ("pairs", 256 * 9216 * 3 // 2), ("pairs", 256 * 4 * 4096 + 1), ("keys", 1 << 25)])
@pytest.mark.parametrize("delta", [-1, 0, 1, 12287])
def test_geometry_switch_points(G, mode, threshold, delta):
    """Sizes right at the small -> large geometry switch of each kernel family (3/2 large tiles per CU on 256 CUs; the
    large tile is 10240 pairs / 16384 keys for the 128-byte-line kernel of 32-bit keys, 8192 pairs for 64-bit keys;
    12288 pairs / 8192 pairs are the switches of the kernel that unaligned arrays fall back to; 32-bit keys with values
    stay on the small geometry while one round of its workgroups takes the input: 256 CUs x 4 x 4096 pairs -- the last of
    those sizes, 2^22, is also the first one sorted with a device-side pass plan; keys-only sorts of 32-bit keys stay on it up to
    2^25 keys, where the line kernel and the attempt to end in LDS begin)."""
    n = threshold + delta
    rng = np.random.default_rng(n)
    if mode == "u64":
        keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    else:
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[rng.integers(0, n, n // 50)] = keys[0]  # some duplicates
    if mode == "keys":
        kb = G.ShaderStorageBuffer(keys)
        G.RadixSort().sort_keys(kb, n)
        assert (kb.get_data(np.uint32) == np.sort(keys, kind="stable")).all()
        return
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, keys, vals, key_bytes=8 if mode == "u64" else 4)
    ek, ev = O.stable_sort_pairs(keys, vals) if mode == "pairs" else (None, None)
    if mode == "pairs":
        assert (gk == ek).all() and (gv == ev).all()
    else:
        order = np.argsort(keys, kind="stable")
        assert (gk == keys[order]).all() and (gv == vals[order]).all()


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("n", [16385, 20000, 4096 * 5 + 1, 65536, 4096 * 32, 4096 * 32 + 1, 2048 * 32, 2048 * 33])
def test_fused_row_scan_path(G, bits, n, monkeypatch):
    """Up to 32 workgroups the scatter kernel sums the count table itself (no row-scan launch); the three-launch path
    (GLU_HIP_SORT_NO_FUSED_SCAN=1) must give the same result, for pairs, keys only and 64-bit keys."""
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::7] = keys[3]
    vals = np.arange(n, dtype=np.uint32)
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    ek, ev = O.stable_sort_pairs(keys, vals)
    o64 = np.argsort(k64, kind="stable")
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("GLU_HIP_SORT_NO_FUSED_SCAN", env)
        gk, gv = gpu_sort(G, keys, vals, bits=bits)
        assert (gk == ek).all() and (gv == ev).all()
        kb = G.ShaderStorageBuffer(keys)
        G.RadixSort(digit_bits=bits).sort_keys(kb, n)
        assert (kb.get_data(np.uint32) == ek).all()
        gk, gv = gpu_sort(G, k64, vals, bits=bits, key_bytes=8)
        assert (gk == k64[o64]).all() and (gv == vals[o64]).all()


def _constant_byte_keys(rng, n, dtype, const_bytes):
    """Random keys whose bytes listed in const_bytes hold one value for the whole array."""
    bits = np.dtype(dtype).itemsize * 8
    keys = rng.integers(0, 2**bits, n, dtype=dtype)
    for b in const_bytes:
        m = dtype(0xFF) << dtype(8 * b)
        keys = (keys & ~m) | (dtype(int(rng.integers(0, 256))) << dtype(8 * b))
    return keys


@pytest.mark.parametrize("bits", DIGIT_BITS)
@pytest.mark.parametrize("const_bytes", [(), (3,), (0,), (1, 2), (0, 1, 2, 3), (2, 3), (0, 3)])
def test_planned_sort_skips_constant_digit_passes(G, bits, const_bytes, monkeypatch):
    """From 2^22 elements up a pass whose digit is the same in every key is skipped on the device (its scatter returns at
    once, the arrays' roles for the later passes follow a device-side plan, an odd number of executed passes is copied
    home at the end).  Every combination of constant bytes must give the stable sort, with and without the plan."""
    n = (1 << 22) + 4321
    rng = np.random.default_rng(len(const_bytes) * 10 + sum(const_bytes) + bits)
    keys = _constant_byte_keys(rng, n, np.uint32, const_bytes)
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("GLU_HIP_SORT_NO_PLAN", env)
        gk, gv = gpu_sort(G, keys, vals, bits=bits)
        assert (gk == keys[order]).all() and (gv == vals[order]).all(), (const_bytes, env)
        kb = G.ShaderStorageBuffer(keys)
        G.RadixSort(digit_bits=bits).sort_keys(kb, n)
        assert (kb.get_data(np.uint32) == keys[order]).all()


@pytest.mark.parametrize("const_bytes", [(7, 6, 5), (4,), (0, 7), (1, 3, 5, 7)])
def test_planned_sort_u64_and_typed(G, const_bytes):
    n = (1 << 22) + 99
    rng = np.random.default_rng(sum(const_bytes))
    k64 = _constant_byte_keys(rng, n, np.uint64, const_bytes)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = gpu_sort(G, k64, vals, key_bytes=8)
    o = np.argsort(k64, kind="stable")
    assert (gk == k64[o]).all() and (gv == vals[o]).all()
    # typed: floats in [1, 2) share sign and exponent (top 9 bits constant); the encode / decode passes still run
    f = (1.0 + rng.random(n)).astype(np.float32)
    kb, vb = G.ShaderStorageBuffer(f), G.ShaderStorageBuffer(vals)
    G.RadixSort().sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, "float32")
    o = np.argsort(f, kind="stable")
    assert (kb.get_data(np.float32) == f[o]).all() and (vb.get_data(np.uint32) == vals[o]).all()
    i64 = (k64 >> np.uint64(40)).astype(np.int64) - (1 << 20)  # small signed range: five constant high bytes after encode? no: sign-extended
    kb = G.ShaderStorageBuffer(i64)
    G.RadixSort().sort_typed_ptr(kb.device_ptr(), None, n, "int64")
    assert (kb.get_data(np.int64) == np.sort(i64)).all()


def test_planned_sort_really_skips(G, monkeypatch):
    """All-equal keys (the reference README's benchmark input): with the plan every scatter returns at once."""
    n = 1 << 24
    keys = np.full(n, 0x12345678, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    times = {}
    for env in ("0", "1"):
        monkeypatch.setenv("GLU_HIP_SORT_NO_PLAN", env)
        s = G.RadixSort()
        s.prepare_internal_buffers(n)
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        s(kb, vb, n)  # warm-up
        s.set_profiling(True)
        s(kb, vb, n)
        G.synchronize()
        times[env] = s.read_profile()["scatter_ms"]
        assert (vb.get_data(np.uint32) == vals).all() and (kb.get_data(np.uint32) == keys).all()
    assert times["0"] < 0.25 * times["1"], times


def test_planned_sort_inside_a_captured_graph(G):
    """The skip decisions live on the device, so one captured graph of a large sort replays correctly on inputs that skip
    different passes (none / the top byte / all of them)."""
    import torch

    n = (1 << 22) + 77
    sorter = G.RadixSort()
    sorter.prepare_internal_buffers(n)
    kt = torch.empty(n, dtype=torch.int32, device="cuda")
    vt = torch.empty(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    rng = np.random.default_rng(2024)
    vals = np.arange(n, dtype=np.uint32)
    inputs = [rng.integers(0, 2**32, n, dtype=np.uint32), rng.integers(0, 2**24, n, dtype=np.uint32),
              np.full(n, 7, dtype=np.uint32), rng.integers(0, 2**32, n, dtype=np.uint32) & np.uint32(0xFF00FF00)]
    with torch.cuda.stream(side):
        kt.copy_(torch.from_numpy(inputs[0].view(np.int32)))
        vt.copy_(torch.from_numpy(vals.view(np.int32)))
        sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, side.cuda_stream)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)
        for keys in inputs:
            kt.copy_(torch.from_numpy(keys.view(np.int32)))
            vt.copy_(torch.from_numpy(vals.view(np.int32)))
            graph.replay()
            side.synchronize()
            order = np.argsort(keys, kind="stable")
            assert (kt.cpu().numpy().view(np.uint32) == keys[order]).all() and (vt.cpu().numpy().view(np.uint32) == vals[order]).all()


# ---- paired passes (radix_pair_passes.hpp): sorts of >= 2^22 elements with 8-bit digits read the keys once per PAIR of
# passes; the second pass of a pair ("follower") takes its count table from the two-digit histogram of the first, unless
# the data sends it back to its own count kernel.  glu_radix_sort_read_plan says what happened.

def _sort_and_plan(G, keys, vals, passes, key_bytes=4, env=None, run=None, bits=8):
    env = dict(env or {})
    env.setdefault("GLU_HIP_SORT_PAIR_MIN", "1")  # pair from the smallest planned sort up (default: from 2^28 bytes of keys)
    sorter = G.RadixSort(digit_bits=bits, options=env)  # (switches set on the object: glu_radix_sort_set_option)
    kb = G.ShaderStorageBuffer(keys)
    vb = G.ShaderStorageBuffer(vals) if vals is not None else None
    if run is not None:
        run(sorter, kb, vb)
    elif vals is None:
        sorter.sort_keys_ptr(kb.device_ptr(), keys.size, 0, None, key_bytes=key_bytes)
    else:
        sorter(kb, vb, keys.size, 0, key_bytes=key_bytes)
    G.synchronize()
    skipped, alone, roles = sorter.read_plan(passes, roles=True)
    _sort_and_plan.last_roles = roles
    _sort_and_plan.last_skip_raw = skipped  # 1 = identity found by counting, 2 = known before counting
    skipped = [1 if x else 0 for x in skipped]
    return kb.get_data(keys.dtype), (vb.get_data(np.uint32) if vb is not None else None), skipped, alone


PAIR_N = (1 << 24) + 4321


def _check_against_oracle(keys, vals, gk, gv):
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()


def test_paired_passes_uniform_keys_take_every_table_from_the_leader(G):
    rng = np.random.default_rng(31)
    keys = rng.integers(0, 2**32, PAIR_N, dtype=np.uint32)
    vals = np.arange(PAIR_N, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4)
    _check_against_oracle(keys, vals, gk, gv)
    assert skipped == [0, 0, 0, 0] and alone == [0, 0, 0, 0] and _sort_and_plan.last_roles == [1, 2, 1, 2]
    # and with the switch off the same result from four counting kernels
    gk2, gv2, _, alone2 = _sort_and_plan(G, keys, vals, 4, env={"GLU_HIP_SORT_PAIRS": "0"})
    assert (gk2 == gk).all() and (gv2 == gv).all() and alone2 == [0, 0, 0, 0] and _sort_and_plan.last_roles == [0, 0, 0, 0]
    # by default a sort of this size (2^26 bytes of keys) does not pair either: the tables would cost more than the keys
    gk3, gv3, _, _ = _sort_and_plan(G, keys, vals, 4, env={"GLU_HIP_SORT_PAIR_MIN": "0"})
    assert (gk3 == gk).all() and (gv3 == gv).all() and _sort_and_plan.last_roles == [0, 0, 0, 0]


def test_paired_passes_common_digit_value_sends_the_follower_back_to_counting(G):
    """10 % of the keys share one low byte: the leader's units of that digit value are longer than 1/16 of a workgroup's
    share, runs of whole units cannot be balanced, pass 1 counts for itself; the second pair is unaffected."""
    rng = np.random.default_rng(32)
    keys = rng.integers(0, 2**32, PAIR_N, dtype=np.uint32)
    hot = rng.random(PAIR_N) < 0.10
    keys[hot] = (keys[hot] & np.uint32(0xFFFFFF00)) | np.uint32(0x77)
    vals = np.arange(PAIR_N, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4)
    _check_against_oracle(keys, vals, gk, gv)
    assert alone == [0, 1, 0, 0] and skipped == [0, 0, 0, 0]


def test_paired_passes_counter_overflow_is_found(G):
    """The first 200 000 keys share their low 16 bits: the 16-bit counter of that digit pair overflows in the first
    workgroups' two-digit histograms.  With the unit-length rule switched off, the row-sum check is what sends pass 1
    back to counting."""
    rng = np.random.default_rng(33)
    keys = rng.integers(0, 2**32, PAIR_N, dtype=np.uint32)
    keys[:200000] = (keys[:200000] & np.uint32(0xFFFF0000)) | np.uint32(0x1234)
    vals = np.arange(PAIR_N, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4, env={"GLU_HIP_SORT_PAIR_UNIT_DIV": "0"})
    _check_against_oracle(keys, vals, gk, gv)
    assert alone == [0, 1, 0, 0]
    # the default rule catches the same input earlier (those units are long)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4)
    _check_against_oracle(keys, vals, gk, gv)
    assert alone == [0, 1, 0, 0]


def test_paired_passes_run_of_many_tiny_units_counts_instead(G):
    """Half of the low-byte values are rare (1 key in 256 has one of them): their 128 x 256 units hold about two keys each
    and one follower workgroup's run would span ten thousands of them."""
    rng = np.random.default_rng(34)
    keys = rng.integers(0, 2**32, PAIR_N, dtype=np.uint32)
    rare = rng.random(PAIR_N) < 1.0 / 256
    keys[~rare] &= np.uint32(0xFFFFFF7F)
    keys[rare] |= np.uint32(0x80)
    vals = np.arange(PAIR_N, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4)
    _check_against_oracle(keys, vals, gk, gv)
    assert alone == [0, 1, 0, 0]


@pytest.mark.parametrize("case", ["sorted", "reverse", "all_equal", "two_values", "low_16_bits_only", "high_16_bits_only",
                                  "byte_1_constant", "few_distinct"])
def test_paired_passes_structured_inputs(G, case):
    n = PAIR_N
    rng = np.random.default_rng(35)
    if case == "sorted":
        keys = np.arange(n, dtype=np.uint32) * np.uint32(251)
    elif case == "reverse":
        keys = (np.uint32(0xFFFFFFFF) - np.arange(n, dtype=np.uint32) * np.uint32(17))
    elif case == "all_equal":
        keys = np.full(n, 0xA5A5A5A5, dtype=np.uint32)
    elif case == "two_values":
        keys = np.where(rng.random(n) < 0.5, np.uint32(0x01020304), np.uint32(0xF1F2F3F4)).astype(np.uint32)
    elif case == "low_16_bits_only":
        keys = rng.integers(0, 2**16, n, dtype=np.uint32)
    elif case == "high_16_bits_only":
        keys = rng.integers(0, 2**16, n, dtype=np.uint32) << np.uint32(16)
    elif case == "byte_1_constant":
        keys = (rng.integers(0, 2**32, n, dtype=np.uint32) & np.uint32(0xFFFF00FF)) | np.uint32(0x4200)
    else:
        keys = rng.integers(0, 40, n, dtype=np.uint32) * np.uint32(0x01010101)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4)
    _check_against_oracle(keys, vals, gk, gv)
    if case == "low_16_bits_only":
        assert skipped == [0, 0, 1, 1] and alone[:2] == [0, 0]
    if case == "high_16_bits_only":
        assert skipped == [1, 1, 0, 0]
    if case == "byte_1_constant":
        assert skipped == [0, 1, 0, 0]
    if case == "all_equal":
        assert skipped == [1, 1, 1, 1]


def test_paired_passes_u64_keys_only_and_bit_ranges(G):
    n = (3 << 21) - 4099  # (below 3 * 2^21: from there 64-bit keys first try to end in LDS, and these are the ordinary passes' tests)
    rng = np.random.default_rng(36)
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    k64[::5] &= np.uint64(0x0000FFFFFFFFFFFF)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, k64, vals, 8, key_bytes=8)
    order = np.argsort(k64, kind="stable")
    assert (gk == k64[order]).all() and (gv == vals[order]).all()
    # (a fifth of the keys has zero top bytes: the last leader's units of digit value 0 are long, pass 7 counts for itself)
    assert alone == [0, 0, 0, 0, 0, 0, 0, 1] and skipped == [0] * 8
    # 64-bit keys whose upper word is zero: the last two pairs of passes are identities
    small = k64 & np.uint64(0xFFFFFFFF)
    gk, gv, skipped, alone = _sort_and_plan(G, small, vals, 8, key_bytes=8)
    order = np.argsort(small, kind="stable")
    assert (gk == small[order]).all() and (gv == vals[order]).all()
    assert skipped == [0, 0, 0, 0, 1, 1, 1, 1]
    # keys only
    gk, _, skipped, alone = _sort_and_plan(G, k64, None, 8, key_bytes=8)
    assert (gk == np.sort(k64)).all() and alone == [0, 0, 0, 0, 0, 0, 0, 1]
    k32 = rng.integers(0, 2**32, n, dtype=np.uint32)
    gk, _, skipped, alone = _sort_and_plan(G, k32, None, 4)
    assert (gk == np.sort(k32)).all() and alone == [0] * 4
    # bit ranges: [4, 28) = three 8-bit passes (a pair and a single), [8, 21) = 8 + 5 bits (a pair with a narrow follower)
    for begin, end, passes in ((4, 28, 3), (8, 21, 2), (0, 12, 2)):
        field = (k32 >> np.uint32(begin)) & np.uint32((1 << (end - begin)) - 1)
        order = np.argsort(field, kind="stable")
        gk, gv, skipped, alone = _sort_and_plan(
            G, k32, vals, passes, run=lambda s, kb, vb: s.sort_bit_range_ptr(kb.device_ptr(), vb.device_ptr(), n, begin, end, None, 4))
        assert (gk == k32[order]).all() and (gv == vals[order]).all(), (begin, end)
        assert alone[:passes] == [0] * passes


@pytest.mark.parametrize("dtype", ["float32", "int32", "float64", "int64"])
def test_paired_passes_typed_keys(G, dtype):
    """Signed / float keys: the first pass encodes on load (it counts alone), the passes after it pair up, the last one
    decodes on store."""
    n = (1 << 22) + 12345
    rng = np.random.default_rng(37)
    dt = np.dtype(dtype)
    if dt.kind == "f":
        keys = (rng.standard_normal(n) * 1e3).astype(dt)
        keys[::50] = 0.0
        keys[1::50] = -0.0
    else:
        info = np.iinfo(dt)
        keys = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
    vals = np.arange(n, dtype=np.uint32)
    passes = dt.itemsize
    gk, gv, skipped, alone = _sort_and_plan(
        G, keys, vals, passes, run=lambda s, kb, vb: s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, dtype))
    u = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
    sign = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
    image = (u ^ sign) if dt.kind == "i" else np.where(u & sign, ~u, u ^ sign)
    order = np.argsort(image, kind="stable")
    assert (gk.view(u.dtype) == u[order]).all() and (gv == vals[order]).all()
    assert skipped == [0] * passes
    assert _sort_and_plan.last_roles == ([0, 1, 2, 0] if passes == 4 else [0, 1, 2, 1, 2, 1, 2, 0])
    if dt.kind == "i":
        assert alone == [0] * passes  # uniform digits: every follower took its table from its leader


@pytest.mark.parametrize("blocks", ["3", "37", "255"])
def test_paired_passes_fewer_workgroups_than_cus(G, blocks):
    """GLU_HIP_SORT_BLOCKS caps the grid (what glu_dist does to leave CUs to RCCL): units, runs and tables follow the grid."""
    rng = np.random.default_rng(38)
    n = (1 << 23) + 5
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::13] = keys[5]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4, env={"GLU_HIP_SORT_BLOCKS": blocks})
    _check_against_oracle(keys, vals, gk, gv)
    assert _sort_and_plan.last_roles == [1, 2, 1, 2]


# ---- the same with the reference's 4-bit digits: units are (digit value, sub-block of 1/16 of a workgroup's block) -------

def test_paired_passes_4bit_uniform_and_switch(G):
    rng = np.random.default_rng(41)
    keys = rng.integers(0, 2**32, PAIR_N, dtype=np.uint32)
    vals = np.arange(PAIR_N, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 8, bits=4)
    _check_against_oracle(keys, vals, gk, gv)
    assert skipped == [0] * 8 and alone == [0] * 8 and _sort_and_plan.last_roles == [1, 2] * 4
    gk2, gv2, _, _ = _sort_and_plan(G, keys, vals, 8, bits=4, env={"GLU_HIP_SORT_PAIRS": "0"})
    assert (gk2 == gk).all() and (gv2 == gv).all() and _sort_and_plan.last_roles == [0] * 8


@pytest.mark.parametrize("case", ["sorted", "reverse", "all_equal", "two_values", "low_16_bits_only", "nibble_2_constant",
                                  "hot_nibble", "rare_nibbles", "duplicates"])
def test_paired_passes_4bit_structured_inputs(G, case):
    n = PAIR_N
    rng = np.random.default_rng(42)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    expect_alone = None
    if case == "sorted":
        keys = np.arange(n, dtype=np.uint32) * np.uint32(251)
    elif case == "reverse":
        keys = (np.uint32(0xFFFFFFFF) - np.arange(n, dtype=np.uint32) * np.uint32(17))
    elif case == "all_equal":
        keys = np.full(n, 0xA5A5A5A5, dtype=np.uint32)
    elif case == "two_values":
        keys = np.where(rng.random(n) < 0.5, np.uint32(0x01020304), np.uint32(0xF1F2F3F4)).astype(np.uint32)
    elif case == "low_16_bits_only":
        keys = rng.integers(0, 2**16, n, dtype=np.uint32)
    elif case == "nibble_2_constant":
        keys = (keys & np.uint32(0xFFFFF0FF)) | np.uint32(0x300)
    elif case == "hot_nibble":  # 60 % of the keys share the lowest nibble: no way back to counting with 4-bit digits
        hot = rng.random(n) < 0.6
        keys[hot] = (keys[hot] & np.uint32(0xFFFFFFF0)) | np.uint32(0x7)
        expect_alone = [0] * 8
    elif case == "rare_nibbles":  # half of the values of the lowest nibble are rare: one run of ten thousands of tiny units
        rare = rng.random(n) < 1.0 / 256
        keys[~rare] &= np.uint32(0xFFFFFFF7)
        keys[rare] |= np.uint32(0x8)
        expect_alone = [0, 1, 0, 0, 0, 0, 0, 0]
    else:
        keys = np.repeat(keys[: n // 64 + 1], 64)[:n].copy()
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 8, bits=4)
    _check_against_oracle(keys, vals, gk, gv)
    assert _sort_and_plan.last_roles == [1, 2] * 4
    if expect_alone is not None:
        assert alone == expect_alone
    if case == "low_16_bits_only":
        assert skipped == [0, 0, 0, 0, 1, 1, 1, 1]
    if case == "nibble_2_constant":
        assert skipped == [0, 0, 1, 0, 0, 0, 0, 0]
    if case == "all_equal":
        assert skipped == [1] * 8


def test_paired_passes_4bit_u64_typed_keys_only_bit_ranges_and_capped_grid(G):
    n = (1 << 23) + 99
    rng = np.random.default_rng(43)
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, k64, vals, 16, key_bytes=8, bits=4)
    order = np.argsort(k64, kind="stable")
    assert (gk == k64[order]).all() and (gv == vals[order]).all()
    assert alone == [0] * 16 and skipped == [0] * 16 and _sort_and_plan.last_roles == [1, 2] * 8
    gk, _, _, _ = _sort_and_plan(G, k64, None, 16, key_bytes=8, bits=4)
    assert (gk == np.sort(k64)).all()
    k32 = rng.integers(0, 2**32, n, dtype=np.uint32)
    gk, _, _, alone = _sort_and_plan(G, k32, None, 8, bits=4)
    assert (gk == np.sort(k32)).all() and alone == [0] * 8
    # float keys: the first pass encodes on load and stands alone, the last one decodes on store
    f = (rng.standard_normal(n) * 1e3).astype(np.float32)
    gk, gv, _, _ = _sort_and_plan(G, f, vals, 8, bits=4,
                                  run=lambda s, kb, vb: s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, "float32"))
    u = f.view(np.uint32)
    image = np.where(u & np.uint32(0x80000000), ~u, u ^ np.uint32(0x80000000))
    order = np.argsort(image, kind="stable")
    assert (gk.view(np.uint32) == u[order]).all() and (gv == vals[order]).all()
    assert _sort_and_plan.last_roles == [0, 1, 2, 1, 2, 1, 2, 0]
    # bit ranges: [3, 17) = 4 + 4 + 4 + 2 bits
    for begin, end, passes in ((3, 17, 4), (0, 7, 2), (20, 32, 3)):
        field = (k32 >> np.uint32(begin)) & np.uint32((1 << (end - begin)) - 1)
        order = np.argsort(field, kind="stable")
        gk, gv, _, _ = _sort_and_plan(
            G, k32, vals, passes, bits=4, run=lambda s, kb, vb: s.sort_bit_range_ptr(kb.device_ptr(), vb.device_ptr(), n, begin, end, None, 4))
        assert (gk == k32[order]).all() and (gv == vals[order]).all(), (begin, end)
    for blocks in ("3", "37"):
        gk, gv, _, _ = _sort_and_plan(G, k32, vals, 8, bits=4, env={"GLU_HIP_SORT_BLOCKS": blocks})
        _check_against_oracle(k32, vals, gk, gv)
        assert _sort_and_plan.last_roles == [1, 2] * 4


@pytest.mark.parametrize("bits", DIGIT_BITS)
def test_passes_on_key_bits_that_do_not_vary_are_identities_without_counting(G, bits):
    """The count kernel of the first pass of an unsigned-key sort notes which key bits vary over the input; a later pass
    whose digit lies in constant bits is skipped without reading the keys (skip value 2), also inside pairs of passes
    and for 64-bit keys that hold 32-bit numbers."""
    n = (1 << 23) + 321
    rng = np.random.default_rng(51 + bits)
    vals = np.arange(n, dtype=np.uint32)
    per = 8 // bits  # passes per key byte
    # bytes 1 and 3 of the key constant (non-zero), bytes 0 and 2 random
    keys = (rng.integers(0, 2**32, n, dtype=np.uint32) & np.uint32(0x00FF00FF)) | np.uint32(0x5A00C300)
    gk, gv, skipped, alone = _sort_and_plan(G, keys, vals, 4 * per, bits=bits)
    _check_against_oracle(keys, vals, gk, gv)
    assert _sort_and_plan.last_skip_raw == [0] * per + [2] * per + [0] * per + [2] * per
    # the switch: the same passes are found to be identities by counting
    gk2, gv2, skipped2, _ = _sort_and_plan(G, keys, vals, 4 * per, bits=bits, env={"GLU_HIP_SORT_NO_BIT_SHORTCUT": "1"})
    assert (gk2 == gk).all() and (gv2 == gv).all() and _sort_and_plan.last_skip_raw == [0] * per + [1] * per + [0] * per + [1] * per
    # a constant FIRST digit is found by counting (nothing is known before the first pass)
    keys0 = (keys & np.uint32(0xFFFFFF00)) | np.uint32(0x11)
    gk, gv, skipped, alone = _sort_and_plan(G, keys0, vals, 4 * per, bits=bits)
    _check_against_oracle(keys0, vals, gk, gv)
    assert _sort_and_plan.last_skip_raw == [1] + [2] * (per - 1) + [2] * per + [0] * per + [2] * per
    # all keys equal: one read of the keys
    same = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    gk, gv, skipped, alone = _sort_and_plan(G, same, vals, 4 * per, bits=bits)
    assert (gk == same).all() and (gv == vals).all()
    assert _sort_and_plan.last_skip_raw == [1] + [2] * (4 * per - 1)
    # 64-bit keys holding 32-bit numbers, keys only (below 3 * 2^21 keys: from there such a sort ends in LDS, its runs taken from bits
    # [16, 32) -- tests/test_gpu_lds_finish.py::test_u64_keys_of_a_smaller_range_on_the_first_sort)
    k64 = rng.integers(0, 2**32, (3 << 21) - 5000, dtype=np.uint64)
    gk, _, skipped, alone = _sort_and_plan(G, k64, None, 8 * per, key_bytes=8, bits=bits)
    assert (gk == np.sort(k64)).all()
    assert _sort_and_plan.last_skip_raw == [0] * (4 * per) + [2] * (4 * per)
    # signed keys: the sort is on encoded bit patterns, nothing is noted, constant bytes are found by counting
    i32 = (keys & np.uint32(0x7FFFFFFF)).view(np.int32)
    gk, gv, skipped, alone = _sort_and_plan(
        G, i32, vals, 4 * per, bits=bits, run=lambda s, kb, vb: s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, "int32"))
    order = np.argsort(i32, kind="stable")
    assert (gk == i32[order]).all() and (gv == vals[order]).all()
    assert 2 not in _sort_and_plan.last_skip_raw


@pytest.mark.parametrize("bits", DIGIT_BITS)
def test_equal_element_shares_per_workgroup(G, bits):
    """GLU_HIP_SORT_EQUAL_SHARES=1 (a tuning switch): the line path cuts the input into equal element shares instead of whole
    tiles, every workgroup ends on a partial tile; with and without paired passes, 32- and 64-bit keys."""
    rng = np.random.default_rng(61 + bits)
    for n, pair_min in ((4_100_000, "0"), ((1 << 23) + 4321, "1"), (6_000_001, "1")):
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        keys[::9] = keys[3]
        vals = np.arange(n, dtype=np.uint32)
        gk, gv, _, _ = _sort_and_plan(G, keys, vals, 32 // bits, bits=bits,
                                      env={"GLU_HIP_SORT_EQUAL_SHARES": "1", "GLU_HIP_SORT_PAIR_MIN": pair_min})
        _check_against_oracle(keys, vals, gk, gv)
    n = 5_000_003
    k64 = rng.integers(0, 2**64, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv, _, _ = _sort_and_plan(G, k64, vals, 64 // bits, key_bytes=8, bits=bits,
                                  env={"GLU_HIP_SORT_EQUAL_SHARES": "1", "GLU_HIP_SORT_PAIR_MIN": "1"})
    order = np.argsort(k64, kind="stable")
    assert (gk == k64[order]).all() and (gv == vals[order]).all()
