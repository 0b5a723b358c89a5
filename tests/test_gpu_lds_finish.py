"""GPU parity tests of the sort that ends in LDS (radix_lds_finish.hpp): a large sort of whole 32-bit keys first tries the two
counting passes on the TOP 16 key bits and one pass that orders every run of equal top bits inside LDS; the device accepts
that only if no run is longer than a workgroup's tile, else the four ordinary passes run.  Either way the result is the
stable sort of glu::RadixSort::operator() (reference glu/RadixSort.hpp:273-334) -- compared bit for bit with the oracle,
through the C ABI."""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return built


def _sorter(G, **options):
    """A sort object with the given switches set on it (glu_radix_sort_set_option; the process environment stays as it is)."""
    return G.RadixSort(options=options)


# small sizes: pair the passes and make the attempt from the smallest planned sort up (defaults: 2^26 elements)
SMALL = dict(GLU_HIP_SORT_PAIR_MIN=1, GLU_HIP_SORT_FINISH_MIN=1)
CAP_SMALL = 1536  # the smallest geometry of the in-LDS pass: 256 threads x 6 pairs


def _run(G, sorter, keys, vals):
    kb = G.ShaderStorageBuffer(keys)
    if vals is None:
        sorter.sort_keys_ptr(kb.device_ptr(), keys.size)
        G.synchronize()
        return kb.get_data(np.uint32), None, sorter.read_finish()
    vb = G.ShaderStorageBuffer(vals)
    sorter(kb, vb, keys.size)
    G.synchronize()
    return kb.get_data(np.uint32), vb.get_data(np.uint32), sorter.read_finish()


def _check(keys, vals, gk, gv):
    if vals is None:
        assert (gk == np.sort(keys, kind="stable")).all()
        return
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all(), "keys differ from the oracle at %s" % np.flatnonzero(gk != ek)[:5]
    assert (gv == ev).all(), "values differ from the oracle at %s" % np.flatnonzero(gv != ev)[:5]


N_SMALL = (1 << 22) + 54321


def _uniform(n, seed):
    return np.random.default_rng(seed).integers(0, 2**32, n, dtype=np.uint32)


def test_uniform_keys_end_in_lds(G):
    keys, vals = _uniform(N_SMALL, 1), np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    gk, gv, fin = _run(G, s, keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["capacity"] == CAP_SMALL
    top = np.bincount(keys >> 16, minlength=65536)
    assert fin["longest_run"] == top.max()
    # the ordinary passes were the sequence not taken: all four known to be skipped before counting
    skipped, alone, roles = s.read_plan(4, roles=True)
    assert skipped == [2, 2, 2, 2] and roles == [1, 2, 1, 2]
    # same object, switch off: the same result from the four ordinary passes
    off = _sorter(G, GLU_HIP_SORT_LDS_FINISH=0, **SMALL)
    gk2, gv2, fin2 = _run(G, off, keys, vals)
    assert (gk2 == gk).all() and (gv2 == gv).all() and fin2["attempted"] == 0


def test_keys_only_end_in_lds(G):
    keys = _uniform((1 << 23) + 4321, 2)
    s = _sorter(G, GLU_HIP_SORT_LARGE_MIN=1, **SMALL)  # (the keys-only line kernel -- and with it the attempt -- starts at 2^25 keys)
    gk, _, fin = _run(G, s, keys, None)
    _check(keys, None, gk, None)
    assert fin["attempted"] == 1 and fin["accepted"] == 1


def test_duplicate_keys_inside_runs_keep_their_order(G):
    """Few distinct low 16 bits: every run is full of equal keys; the values must come out in input order."""
    rng = np.random.default_rng(3)
    keys = (rng.integers(0, 65536, N_SMALL, dtype=np.uint32) << 16) | rng.integers(0, 3, N_SMALL, dtype=np.uint32) * np.uint32(0x0101)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run(G, _sorter(G, **SMALL), keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["accepted"] == 1


def test_small_key_range_under_the_host_side_guess(G):
    """20-bit keys with the round-4 rule for the runs' bits (GLU_HIP_SORT_DEVICE_TOP=0: a fresh object takes the whole key's top 16):
    16 runs of a quarter million pairs each.  Refused: the segmented passes over all of them would move 76 bytes per pair where three
    ordinary passes -- the keys' top byte is constant -- move 60 (the plan's cost rule, round 6; round 5 refused for the share of the
    pairs in long runs)."""
    rng = np.random.default_rng(4)
    keys = rng.integers(0, 1 << 20, N_SMALL, dtype=np.uint32)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0, **SMALL)
    gk, gv, fin = _run(G, s, keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 0 and fin["longest_run"] > CAP_SMALL
    skipped, alone, roles = s.read_plan(4, roles=True)
    assert skipped[:3] == [0, 0, 0] and skipped[3] != 0 and roles == [1, 2, 1, 2]


def test_all_keys_equal_is_refused(G):
    """One run of n equal keys: no key byte varies, so the ordinary passes are all identities and cost nothing -- the plan refuses
    (round 6's cost rule; at 2^28 keys the blocks of the leader's count kernel would not even put their wrapped counters right)."""
    keys = np.full(N_SMALL, 0xDEADBEEF, dtype=np.uint32)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run(G, _sorter(G, **SMALL), keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 0


def _with_one_run_of(n, length, seed, run=0x1234):
    """Uniform keys, except that exactly `length` of them (at random positions) have the top 16 bits `run`."""
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    clash = (keys >> 16) == run
    keys[clash] ^= np.uint32(0x80000000)  # out of the run
    pos = rng.choice(n, size=length, replace=False)
    keys[pos] = (np.uint32(run) << 16) | rng.integers(0, 65536, length, dtype=np.uint32)
    return keys


@pytest.mark.parametrize("length,accepted,capacity", [(CAP_SMALL - 1, 1, 1536), (CAP_SMALL, 1, 1536), (CAP_SMALL + 1, 1, 2560),
                                                      (2560, 1, 2560), (2561, 1, 4608), (4608, 1, 4608), (4609, 0, 4608),
                                                      (4608 + 5000, 0, 4608)])
def test_the_longest_run_decides(G, length, accepted, capacity):
    """Round-4 rule (GLU_HIP_SORT_LONG_RUNS=0): the in-LDS pass is enqueued
    in the tile geometry that suits uniform keys of this count (here 256 x 6 = 1536 pairs) and in the next two larger ones; the
    device runs the smallest whose tile holds the longest run.  A run of exactly a tile's capacity is sorted in that tile, one
    pair more takes the next, and one pair more than the largest enqueued tile sends the sort to the ordinary passes."""
    keys = _with_one_run_of(N_SMALL, length, 5)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run(G, _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, **SMALL), keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == accepted and fin["longest_run"] == length and fin["capacity"] == capacity


# ---- round 5: runs longer than the tile go to two segmented passes over just their elements; the rest of the sort ends in LDS

@pytest.mark.parametrize("length", [CAP_SMALL, CAP_SMALL + 1, 4609, 4608 + 5000, 200_000])
def test_one_long_run_no_longer_refuses_the_sort(G, length):
    keys = _with_one_run_of(N_SMALL, length, 5)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    gk, gv, fin = _run(G, s, keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["longest_run"] == length and fin["capacity"] == CAP_SMALL, fin
    lr = s.read_long_runs()
    assert lr["runs"] == (1 if length > CAP_SMALL else 0) and lr["pairs"] == (length if length > CAP_SMALL else 0), lr


# ---- round 6: the same for every kind of key -- keys only, 64-bit keys, signed and float keys (the long runs' last segmented pass
# decodes what the in-LDS pass, which leaves those runs alone, would have decoded)

def _sort_kind(G, s, kind, keys_u, vals):
    """keys_u: unsigned bit patterns (np.uint32 / np.uint64) of the order the sort must produce for `kind`; returns (keys out as
    unsigned patterns, values out or None, read_finish, read_long_runs)."""
    name = {"u32_pairs": None, "u32_keys_only": None, "u64_pairs": None, "int32": "int32", "float64": "float64", "int64_keys_only": "int64"}[kind]
    with_vals = not kind.endswith("keys_only")
    if name is None:
        stored = keys_u
    else:
        # the bit patterns whose natural order (as int / float) is the unsigned order of keys_u: invert the order-preserving code
        top = keys_u.dtype.type(1) << keys_u.dtype.type(8 * keys_u.itemsize - 1)
        stored = (keys_u ^ top) if name.startswith("int") else np.where(keys_u & top, keys_u ^ top, ~keys_u)
    kb = G.ShaderStorageBuffer(stored)
    vb = G.ShaderStorageBuffer(vals) if with_vals else None
    if name is None and keys_u.itemsize == 4:
        if with_vals:
            s(kb, vb, keys_u.size)
        else:
            s.sort_keys_ptr(kb.device_ptr(), keys_u.size)
    elif name is None:
        s(kb, vb, keys_u.size, 0, key_bytes=8)
    else:
        s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr() if with_vals else None, keys_u.size, name)
    G.synchronize()
    out = kb.get_data(keys_u.dtype)
    if name is not None:
        top = keys_u.dtype.type(1) << keys_u.dtype.type(8 * keys_u.itemsize - 1)
        out = (out ^ top) if name.startswith("int") else np.where(out & top, ~out, out ^ top)
    return out, (vb.get_data(np.uint32) if with_vals else None), s.read_finish(), s.read_long_runs()


KINDS = ["u32_pairs", "u32_keys_only", "u64_pairs", "int32", "float64", "int64_keys_only"]


def _kind_keys(kind, n, seed, long_runs):
    """Uniform unsigned patterns of the kind's width with `long_runs` = [(run value of the top 16 bits, length)] planted."""
    wide = kind in ("u64_pairs", "float64", "int64_keys_only")
    dt = np.uint64 if wide else np.uint32
    bits = 64 if wide else 32
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 2**bits, n, dtype=dt)
    if kind == "float64":  # no NaN patterns in the order code: they would still sort, but numpy's view of them is not comparable
        keys &= ~(dt(0x7FF) << dt(52)) | (dt(0x3FF) << dt(52))
    sh = dt(bits - 16)
    taken = np.zeros(n, dtype=bool)
    for run, length in long_runs:
        clash = ((keys >> sh) == dt(run)) & ~taken
        keys[clash] ^= dt(1) << dt(bits - 2)
        free = np.flatnonzero(~taken)
        pos = rng.choice(free, size=length, replace=False)
        low = rng.integers(0, 2**(bits - 16), length, dtype=dt)
        if kind == "float64":
            low &= ~(dt(0xF) << dt(44))
        keys[pos] = (dt(run) << sh) | low
        taken[pos] = True
    return keys


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("length", [CAP_SMALL + 1, 4608 + 5000, 150_000])
def test_long_runs_of_every_key_kind(G, kind, length):
    n = N_SMALL if not kind.endswith("keys_only") or kind.startswith("int64") else (1 << 23) + 4321
    keys = _kind_keys(kind, n, 5, [(0x1234 if kind != "float64" else 0x4234, length)])
    vals = np.arange(n, dtype=np.uint32)
    opts = dict(SMALL)
    if kind == "u32_keys_only":
        opts["GLU_HIP_SORT_LARGE_MIN"] = 1  # (keys-only sorts of 4-byte keys run the line kernel from 2^25 keys: forced here)
    s = _sorter(G, **opts)
    gk, gv, fin, lr = _sort_kind(G, s, kind, keys, vals)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all(), "keys differ at %s" % np.flatnonzero(gk != keys[order])[:5]
    if gv is not None:
        assert (gv == vals[order]).all(), "values differ at %s" % np.flatnonzero(gv != vals[order])[:5]
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["longest_run"] >= length, fin
    assert lr["runs"] == 1 and lr["pairs"] == fin["longest_run"], lr


@pytest.mark.parametrize("kind", ["u64_pairs", "int32", "u32_keys_only", "float64"])
def test_many_long_runs_and_one_percent_zeros_of_other_key_kinds(G, kind):
    n = N_SMALL if kind != "u32_keys_only" else (1 << 23) + 4321
    rng = np.random.default_rng(9)
    runs = [(int(r), int(l)) for r, l in zip(rng.choice(30000, 40, replace=False) + 1000, rng.integers(1600, 9000, 40))]
    keys = _kind_keys(kind, n, 6, runs)
    keys[rng.choice(n, n // 100, replace=False)] = 0  # one run of a single key value, a percent of the input
    vals = np.arange(n, dtype=np.uint32)
    opts = dict(SMALL)
    if kind == "u32_keys_only":
        opts["GLU_HIP_SORT_LARGE_MIN"] = 1
    s = _sorter(G, **opts)
    for _ in range(2):  # (twice on one object)
        gk, gv, fin, lr = _sort_kind(G, s, kind, keys, vals)
        order = np.argsort(keys, kind="stable")
        assert (gk == keys[order]).all()
        if gv is not None:
            assert (gv == vals[order]).all()
        assert fin["accepted"] == 1 and lr["runs"] >= 41, (fin, lr)


@pytest.mark.parametrize("distinct", [17, 100, 1000, 5000])
@pytest.mark.parametrize("kind", ["u32_pairs", "u64_pairs", "int32"])
def test_few_distinct_values_end_in_lds_and_their_runs_stay_put(G, kind, distinct):
    """Every key is one of `distinct` values (scattered over the key space), a few thousand to a few hundred thousand copies each:
    every run of equal top bits is a long run of ONE key value (or of a few: with 5000 values some share their top 16 bits) -- the
    sort ends in LDS, and the segmented passes find the single-valued runs in order as they stand."""
    wide = kind == "u64_pairs"
    dt = np.uint64 if wide else np.uint32
    rng = np.random.default_rng(1000 + distinct)
    n = N_SMALL
    pool = rng.integers(0, 2**(64 if wide else 32), distinct, dtype=dt)
    keys = pool[rng.integers(0, distinct, n)]
    vals = np.arange(n, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    gk, gv, fin, lr = _sort_kind(G, s, kind, keys, vals)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all() and (gv == vals[order]).all()
    assert fin["attempted"] == 1 and fin["accepted"] == 1, (fin, lr)
    if distinct <= 100:  # (a value's copies outgrow every enqueued tile: each is a long run, and nothing but long runs is left)
        assert lr["runs"] >= distinct * 0.95 and lr["pairs"] == n, (fin, lr)


# ---- round 6: heavy hitters.  A key value that holds more than 6 % of a block of the leader's count kernel wraps a 16-bit counter of the
# two-digit table; round 5 noticed and refused (no exact run lengths).  The block now counts such rows again with 32-bit counters
# (radix_pair_passes.hpp, wide rows): the lengths are exact, the sort ends in LDS, the heavy hitters are long runs of one value.

@pytest.mark.parametrize("shape", ["three_values", "ten_percent_zeros", "half_one_value_u64", "three_values_int32", "all_equal"])
def test_heavy_hitters_that_wrap_the_16_bit_counters(G, shape):
    n = (1 << 26) + 1234  # (256 blocks of 2^18 keys: a value holding a third of them wraps a 16-bit counter in every block)
    rng = np.random.default_rng(606)
    kind = "u32_pairs"
    if shape == "three_values":
        keys = np.array([0x00000000, 0x7F00FF01, 0xFFFFFFFF], dtype=np.uint32)[rng.integers(0, 3, n)]
        want_runs = 3
    elif shape == "ten_percent_zeros":
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        keys[rng.random(n) < 0.35] = 0  # (35 %: 10 % would not wrap at this size, where a block is 2^18 keys)
        want_runs = 1
    elif shape == "half_one_value_u64":
        kind = "u64_pairs"
        keys = rng.integers(0, 2**64, n, dtype=np.uint64)
        keys[rng.random(n) < 0.5] = np.uint64(0x0123456789ABCDEF)
        want_runs = 1
    elif shape == "three_values_int32":
        kind = "int32"
        keys = np.array([0x00000005, 0x80000000, 0xFFFFFFF0], dtype=np.uint32)[rng.integers(0, 3, n)]
        want_runs = 3
    else:
        keys = np.full(n, 0x00C0FFEE, dtype=np.uint32)
        want_runs = 0
    vals = np.arange(n, dtype=np.uint32)
    s = _sorter(G)
    gk, gv, fin, lr = _sort_kind(G, s, kind, keys, vals)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all(), np.flatnonzero(gk != keys[order])[:5]
    assert (gv == vals[order]).all(), np.flatnonzero(gv != vals[order])[:5]
    if shape == "all_equal":
        # (blocks of ONE key value do not put their row right: refused, and the ordinary passes skip on the collected bits)
        assert fin["attempted"] == 1 and fin["accepted"] == 0, fin
    else:
        assert fin["attempted"] == 1 and fin["accepted"] == 1 and lr["runs"] >= want_runs, (fin, lr)
        assert fin["longest_run"] == int(np.bincount((keys >> (keys.dtype.type(8 * keys.itemsize - 16))).astype(np.int64), minlength=65536).max()), fin


@pytest.mark.parametrize("shape", ["long_run_first", "long_run_last", "many_long_runs", "long_runs_of_equal_keys", "zeros_1_percent",
                                   "an_eighth_in_long_runs", "more_than_half_in_long_runs", "too_many_long_runs"])
def test_mixed_long_and_short_runs(G, shape):
    """Long runs at index 0 and 65535, hundreds of them, long runs of one key value (left where they are: their sub-blocks are emptied
    by the first segmented pass's scan kernel), more than half of the pairs in long runs (accepted since round 6), and the way out
    that is left: more than 8192 runs longer than every enqueued tile send the sort to the ordinary passes."""
    rng = np.random.default_rng(77)
    n = N_SMALL  # mean run 64; the tile for uniform keys holds 1536, the largest enqueued 4608
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    want = {"accepted": 1}
    if shape == "long_run_first":
        keys[rng.choice(n, 7000, replace=False)] &= np.uint32(0x0000FFFF)
    elif shape == "long_run_last":
        keys[rng.choice(n, 9001, replace=False)] |= np.uint32(0xFFFF0000)
    elif shape == "many_long_runs":
        pos = rng.choice(n, 200 * 2000, replace=False)  # (a tenth of the pairs: below the eighth that would move the sort to a larger tile)
        keys[pos] = (rng.integers(0, 200, pos.size, dtype=np.uint32) * np.uint32(97) << np.uint32(16)) | (keys[pos] & np.uint32(0xFFFF))
    elif shape == "long_runs_of_equal_keys":
        pos = rng.choice(n, 40 * 5000, replace=False)
        keys[pos] = (rng.integers(0, 40, pos.size, dtype=np.uint32) * np.uint32(0x01010101)) | np.uint32(0x00100000)
    elif shape == "zeros_1_percent":
        keys[rng.random(n) < 0.01] = 0
    elif shape == "an_eighth_in_long_runs":
        pos = rng.choice(n, n // 9, replace=False)
        keys[pos] = (rng.integers(0, 30, pos.size, dtype=np.uint32) << np.uint32(16)) | (keys[pos] & np.uint32(0xFFFF))
    elif shape == "more_than_half_in_long_runs":
        pos = rng.choice(n, n * 6 // 10, replace=False)
        keys[pos] = (rng.integers(0, 30, pos.size, dtype=np.uint32) << np.uint32(16)) | (keys[pos] & np.uint32(0xFFFF))
        # (round 5 refused this; round 6: two segmented passes over the long runs move no more than the ordinary passes would)
    else:  # too_many_long_runs: 9000 runs of 1600 pairs need more than three segments' worth of ... 14.4 M pairs: a larger input
        n = 9000 * 1600 + (1 << 22)
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        keys[: 9000 * 1600] = (np.repeat(np.arange(9000, dtype=np.uint32), 1600) * np.uint32(7) << np.uint32(16)) | (keys[: 9000 * 1600] & np.uint32(0xFFFF))
        keys = keys[rng.permutation(n)]
    vals = np.arange(keys.size, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    gk, gv, fin = _run(G, s, keys, vals)
    _check(keys, vals, gk, gv)
    lengths = np.bincount((keys >> 16).astype(np.int64), minlength=65536)
    lr = s.read_long_runs()
    if shape == "too_many_long_runs":
        # (the uniform-keys tile of this count is 1536 too: 9000 runs outgrow it, and the next tile takes them whole)
        assert fin["accepted"] == 1 and fin["capacity"] == 2560 and lr["runs"] == int((lengths > 2560).sum()), (fin, lr)
        return
    assert fin["attempted"] == 1 and fin["accepted"] == want["accepted"], fin
    if want["accepted"]:
        cap = fin["capacity"]
        assert lr["runs"] == int((lengths > cap).sum()) and lr["pairs"] == int(lengths[lengths > cap].sum()), (fin, lr)
        assert lr["runs"] > 0
    else:
        assert lr["runs"] == 0


def test_long_runs_on_one_object_back_to_back(G):
    """accepted with long runs, accepted without, refused, with long runs again: the descriptors of one sort are rewritten by the
    next in stream order, nothing of a sort survives into the next"""
    rng = np.random.default_rng(78)
    s = _sorter(G, GLU_HIP_SORT_FINISH_BACKOFF=0, **SMALL)
    n = N_SMALL
    for it, kind in enumerate(["long", "plain", "refused", "long", "long"]):
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        if kind == "long":
            keys[rng.choice(n, 3000 + 4000 * it, replace=False)] &= np.uint32(0x0000FFFF)  # run 0 gets them
        elif kind == "refused":  # (full-range keys, six in ten crowded into thirty runs, and -- for this sort -- the round-4 rule: a
            # run longer than every enqueued tile refuses the sort.  By default the long-run passes have taken such inputs since round 6.)
            crowd = rng.choice(n, n * 6 // 10, replace=False)
            keys[crowd] = (rng.integers(0, 30, crowd.size, dtype=np.uint32) * np.uint32(2001) << np.uint32(16)) | (keys[crowd] & np.uint32(0xFFFF))
        s.set_option("SORT_LONG_RUNS", 0 if kind == "refused" else 1)
        vals = np.arange(n, dtype=np.uint32)
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        lr = s.read_long_runs()
        assert fin["accepted"] == (0 if kind == "refused" else 1), (kind, fin)
        assert (lr["runs"] > 0) == (kind == "long"), (kind, lr)


def test_crowded_runs_in_a_tile_the_object_did_not_expect(G):
    """An object whose last sort took the large tile expects it again; the next sort's runs fit the small one, and some of them are
    crowded (four distinct keys): the small tile's kernel, launched with 8192 looping workgroups, lists them and the launch of
    the ballot rounds for 'any tile but the expected one' takes them (found by test_u64_keys_of_a_smaller_range in round 6:
    listed runs that no launch took came out unsorted)."""
    rng = np.random.default_rng(123)
    n = N_SMALL
    vals = np.arange(n, dtype=np.uint32)
    s = _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)  # (the longest run decides the tile)
    first = _with_one_run_of(n, 4000, 31)  # a run of 4000: the 4608 tile
    gk, gv, fin = _run(G, s, first, vals)
    _check(first, vals, gk, gv)
    assert fin["accepted"] == 1 and fin["capacity"] == 4608
    keys = _uniform(n, 32)
    for run in (5, 600, 40000, 65535):  # a few runs of four distinct keys, 1200 pairs each: crowded, and short of the 1536 tile
        clash = (keys >> 16) == run
        keys[clash] ^= np.uint32(0x00010000)
        pos = rng.choice(n, 1200, replace=False)
        keys[pos] = (np.uint32(run) << np.uint32(16)) | (rng.integers(0, 4, 1200, dtype=np.uint32) * np.uint32(0x1111))
    for _ in range(2):
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        assert fin["accepted"] == 1 and fin["capacity"] == 1536, fin


def test_empty_runs_and_runs_of_one(G):
    """Top 16 bits only even, and a handful of runs with a single pair."""
    rng = np.random.default_rng(6)
    keys = rng.integers(0, 2**32, N_SMALL, dtype=np.uint32) & np.uint32(0xFFFEFFFF)
    keys[keys >> 16 == 0x0002] |= np.uint32(0x00040000)  # empty run 2 ...
    keys[7] = 0x00020007                                  # ... but for one pair
    keys[N_SMALL - 1] = 0x00030001                        # an odd run with one pair (the last element)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run(G, _sorter(G, **SMALL), keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["accepted"] == 1


@pytest.mark.parametrize("shape", ["sorted", "reversed", "sorted_runs_reversed_inside"])
def test_presorted_inputs(G, shape):
    keys = np.sort(_uniform(N_SMALL, 7))
    if shape == "reversed":
        keys = keys[::-1].copy()
    elif shape == "sorted_runs_reversed_inside":
        keys = (keys & np.uint32(0xFFFF0000)) | (~keys & np.uint32(0xFFFF))
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run(G, _sorter(G, **SMALL), keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["accepted"] == 1


def test_one_object_alternates_between_the_two_sequences(G):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  The plan is per sort.  An object's first sort takes its runs from the whole key's top 16 bits; later ones from the top 16
    of the bits that varied in the attempt before (a guess the device checks): uniform keys end in LDS, 18-bit keys are refused
    under the first assumption and end in LDS under the second (runs = bits [2, 18)), full-range keys are then refused once --
    bits above 18 vary -- and end in LDS again.  A small sort (no plan) in between changes nothing."""
    # (GLU_HIP_SORT_LONG_RUNS=0: with the long-run passes, the default, the four long runs of 18-bit keys under the first assumption
    # end in LDS too -- test_small_key_range_under_the_host_side_guess)
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0, GLU_HIP_SORT_FINISH_BACKOFF=0, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    steps = [(8, False, 1, 32), (9, True, 0, 32), (10, True, 1, 18), (11, False, 0, 18), (12, False, 1, 32), (13, False, 1, 32)]
    for seed, small_range, accepted, top in steps:
        keys = _uniform(N_SMALL, seed)
        if small_range:
            keys >>= np.uint32(14)
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        assert fin["attempted"] == 1 and fin["accepted"] == accepted and fin["top_bit"] == top, (seed, fin)
        k2 = _uniform(5000, seed)
        gk2, gv2, fin2 = _run(G, s, k2, np.arange(5000, dtype=np.uint32))
        _check(k2, np.arange(5000, dtype=np.uint32), gk2, gv2)
        assert fin2["attempted"] == 0


@pytest.mark.parametrize("bits,garbage", [(28, 0), (28, 0xA0000000), (21, 0), (17, 0x00FE0000), (16, 0), (31, 0)])
def test_keys_of_a_smaller_range_with_the_host_side_guess(G, bits, garbage):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  Keys below 2^bits (with or without constant bits above) crowd into few runs of the whole key's top bits: the first sort is
    refused; it has noted which key bits vary, and the second takes its runs from the top 16 of those -- bits [bits - 16, bits)
    -- and orders the remaining low bits (12, 5, 1, none, 15) inside LDS."""
    s = _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, GLU_HIP_SORT_DEVICE_TOP=0, **SMALL)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    for rep in range(3):
        keys = (_uniform(N_SMALL, 50 + rep) >> np.uint32(32 - bits)) | np.uint32(garbage)
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        if rep == 0:
            # (at this size 28 and 31 bits still fit a tile under the first assumption: 4096 runs of 1024, 32768 of 128)
            assert fin["attempted"] == 1 and fin["accepted"] == (1 if bits >= 28 else 0) and fin["top_bit"] == 32
        else:
            assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["top_bit"] == max(bits, 16), (rep, fin)


def test_u64_keys_of_a_smaller_range(G):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  64-bit keys of 45 varying bits: a digit must stay inside one 32-bit key word, so the runs come from bits [32, 48); of 40
    bits: from [24, 40)."""
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0, **SMALL)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    for bits, top in [(45, 48), (40, 40), (64, 64)]:
        for rep in range(2):
            keys = _uniform64(N_SMALL, 60 + rep) >> np.uint64(64 - bits)
            gk, gv, fin = _run64(G, s, keys, vals)
            ek, ev = O.stable_sort_pairs(keys, vals)
            assert (gk == ek).all() and (gv == ev).all()
            if rep == 1:
                assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["top_bit"] == top, (bits, fin)


def _crowded_and_wide():
    narrow = _uniform(N_SMALL, 13)  # (full-range keys, but six in ten crowd into thirty runs: beyond what the long-run passes take)
    crowd = np.random.default_rng(13).choice(N_SMALL, N_SMALL * 6 // 10, replace=False)
    narrow[crowd] = (np.random.default_rng(14).integers(0, 30, crowd.size, dtype=np.uint32) * np.uint32(2001) << np.uint32(16)) | (narrow[crowd] & np.uint32(0xFFFF))
    return narrow, _uniform(N_SMALL, 14)


def test_every_sort_asks_whatever_the_object_sorted_before(G):
    """What a sort costs does not depend on the object's history: after refused attempts (crowded keys, all-zero keys) the next
    sort of keys that fit ends in LDS at once.  (GLU_HIP_SORT_LONG_RUNS=0: the round-4 rule makes the crowded keys a refusal; by
    default their long runs go to the segmented passes.)"""
    s = _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    narrow, wide = _crowded_and_wide()
    zeros = np.zeros(N_SMALL, dtype=np.uint32)
    seen = []
    for keys in (narrow, narrow, wide, zeros, zeros, wide, narrow, wide):
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        seen.append((fin["attempted"], fin["accepted"]))
    assert seen == [(1, 0), (1, 0), (1, 1), (1, 0), (1, 0), (1, 1), (1, 0), (1, 1)], seen


def test_the_back_off_switch_skips_attempts_after_a_refusal(G):
    """GLU_HIP_SORT_FINISH_BACKOFF=8 (round 4's default): an object whose last attempt was refused skips the next eight attempts,
    then asks again (inputs that never fit pay for the refused attempt's read of the keys once in nine sorts)."""
    s = _sorter(G, GLU_HIP_SORT_FINISH_BACKOFF=8, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)  # (the round-4 rule: crowded keys are refused)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    narrow, wide = _crowded_and_wide()
    seen = []
    for i in range(11):
        keys = narrow if i < 10 else wide
        gk, gv, fin = _run(G, s, keys, vals)
        if i in (0, 1, 9, 10):
            _check(keys, vals, gk, gv)
        seen.append((fin["attempted"], fin["accepted"]))
    assert seen == [(1, 0)] + [(0, 0)] * 8 + [(1, 0)] + [(0, 0)], seen
    # eight more sorts and the object asks again -- this time the keys fit
    for i in range(7):
        _run(G, s, wide, vals)
    gk, gv, fin = _run(G, s, wide, vals)
    _check(wide, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 1


def test_sizes_around_the_geometries_of_the_in_lds_pass(G):
    """Sizes whose mean run length falls into each tile geometry (6 / 10 / 18 pairs per thread), keys-only to keep the host
    side short; sortedness + the multiset (a checksum) instead of a full oracle sort at the largest."""
    import torch

    s = _sorter(G, GLU_HIP_SORT_PAIR_MIN=1, GLU_HIP_SORT_FINISH_MIN=1)
    for n, cap in [((1 << 26) - 77, 1536), ((1 << 27) + 4099, 2560), ((1 << 28) - 3, 4608)]:
        g = torch.Generator(device="cuda:0")
        g.manual_seed(n)
        keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0", generator=g)
        vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
        k0 = keys.clone()
        torch.cuda.synchronize()
        s.run_ptr(keys.data_ptr(), vals.data_ptr(), n)
        G.synchronize()
        fin = s.read_finish()
        assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["capacity"] == cap, fin
        # unsigned order == order of (key ^ 0x80000000) as int32
        flipped = keys ^ torch.tensor(-2**31, dtype=torch.int32, device="cuda:0")
        assert bool((flipped[1:] >= flipped[:-1]).all()), "not sorted"
        # the permutation is a permutation, and carries each key with its value
        assert bool((k0[vals.long()] == keys).all()), "a value does not point at its key"
        # stability: equal neighbours keep ascending values
        eq = keys[1:] == keys[:-1]
        assert bool((vals[1:][eq] > vals[:-1][eq]).all()), "equal keys out of input order"
        del keys, vals, k0, flipped, eq
        torch.cuda.empty_cache()


def test_keys_that_leave_runs_empty_take_a_larger_tile(G):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  31-bit keys (what the reference's test generator draws) fill half of the runs, each twice as long as uniform keys of the
    same count would: the device takes the next tile geometry; 2^28 of them (runs of 8192) and 2^29 full-range keys end in
    the largest tile, 512 threads x 18."""
    import torch

    for n, shift, cap in [((1 << 27) + 333, 1, 4608), ((1 << 28) - 5, 1, 9216), (1 << 29, 0, 9216)]:
        s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0)  # (a fresh object: its first sort takes the runs from the whole key's top bits)
        keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0")
        if shift:
            keys = (keys >> 1) & torch.tensor(0x7FFFFFFF, dtype=torch.int32, device="cuda:0")
        vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
        k0 = keys.clone()
        torch.cuda.synchronize()
        s.run_ptr(keys.data_ptr(), vals.data_ptr(), n)
        G.synchronize()
        fin = s.read_finish()
        assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["capacity"] == cap, (n, fin)
        flipped = keys ^ torch.tensor(-2**31, dtype=torch.int32, device="cuda:0")
        assert bool((flipped[1:] >= flipped[:-1]).all()), "not sorted"
        del flipped
        assert bool((k0[vals.long()] == keys).all()), "a value does not point at its key"
        eq = keys[1:] == keys[:-1]
        assert bool((vals[1:][eq] > vals[:-1][eq]).all()), "equal keys out of input order"
        del keys, vals, k0, eq
        torch.cuda.empty_cache()


def test_31_bit_keys_at_full_size_go_back_to_the_small_tile(G):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  2^28 pairs of 31-bit keys: the first sort ends in LDS in the largest tile (32768 runs of 8192 under the whole key's top
    bits), the second in the tile for uniform keys (65536 runs of 4096 of bits [15, 31))."""
    import torch

    n = 1 << 28
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0)
    mask = torch.tensor(0x7FFFFFFF, dtype=torch.int32, device="cuda:0")
    for rep, (cap, top) in enumerate([(9216, 32), (4608, 31), (4608, 31)]):
        keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0") & mask
        vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()  # (the sort runs on the library's queue, not on torch's stream)
        s.run_ptr(keys.data_ptr(), vals.data_ptr(), n)
        G.synchronize()
        fin = s.read_finish()
        assert fin["accepted"] == 1 and fin["capacity"] == cap and fin["top_bit"] == top, (rep, fin)
        assert bool((keys[1:] >= keys[:-1]).all()), "not sorted"  # (non-negative as int32)
        eq = keys[1:] == keys[:-1]
        assert bool((vals[1:][eq] > vals[:-1][eq]).all()), "equal keys out of input order"
        del keys, vals, eq
        torch.cuda.empty_cache()


def test_beyond_the_largest_geometry_no_attempt_is_made(G):
    import torch

    n = (1 << 29) + (1 << 26)
    s = _sorter(G)
    keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0")
    vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    s.run_ptr(keys.data_ptr(), vals.data_ptr(), n)
    G.synchronize()
    assert s.read_finish()["attempted"] == 0
    flipped = keys ^ torch.tensor(-2**31, dtype=torch.int32, device="cuda:0")
    assert bool((flipped[1:] >= flipped[:-1]).all())


def test_profile_books_the_sequence_that_ran(G):
    """read_profile counts the passes that did the work: two counting passes + the in-LDS pass when accepted, the four
    ordinary passes when refused (GLU_HIP_SORT_LONG_RUNS=0: the crowded keys below are a refusal under the round-4 rule)."""
    s = _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)
    s.set_profiling(True)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    _run(G, s, _uniform(N_SMALL, 11), vals)
    prof = s.read_profile()
    assert prof["passes"] == 2 and prof["finish_passes"] == 1 and prof["finish_ms"] > 0 and prof["scatter_ms"] > 0
    crowd = _uniform(N_SMALL, 12)  # (six keys in ten crowded into thirty runs: refused)
    pos = np.random.default_rng(12).choice(N_SMALL, N_SMALL * 6 // 10, replace=False)
    crowd[pos] = (np.random.default_rng(13).integers(0, 30, pos.size, dtype=np.uint32) * np.uint32(2001) << np.uint32(16)) | (crowd[pos] & np.uint32(0xFFFF))
    _, _, fin = _run(G, s, crowd, vals)
    assert fin["accepted"] == 0
    prof = s.read_profile()
    assert prof["passes"] == 4 and prof["finish_passes"] == 0 and prof["finish_ms"] == 0


def test_partial_sorts_do_not_attempt(G):
    """Sorts of fewer than all key bits keep the ordinary passes."""
    import torch

    s = _sorter(G, **SMALL)
    n = N_SMALL
    keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda:0")
    vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    s.run_ptr(keys.data_ptr(), vals.data_ptr(), n, num_steps=6)
    G.synchronize()
    assert s.read_finish()["attempted"] == 0


def test_one_captured_graph_serves_every_outcome(G):
    """Which sequence of passes runs, and in which tile the in-LDS pass, is decided on the device: one captured graph of a
    sort replays correctly on keys that end in LDS, with and without long runs, of the full and of a smaller range, and on keys
    that are refused."""
    import torch

    n = N_SMALL
    sorter = _sorter(G, GLU_HIP_SORT_FINISH_BACKOFF=0, **SMALL)
    sorter.prepare_internal_buffers(n)
    kt = torch.empty(n, dtype=torch.int32, device="cuda")
    vt = torch.empty(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    vals = np.arange(n, dtype=np.uint32)
    crowd = _uniform(n, 26)  # six keys in ten crowded into thirty runs: thirty long runs
    pos = np.random.default_rng(26).choice(n, n * 6 // 10, replace=False)
    crowd[pos] = (np.random.default_rng(27).integers(0, 30, pos.size, dtype=np.uint32) * np.uint32(2001) << np.uint32(16)) | (crowd[pos] & np.uint32(0xFFFF))
    # (a long run goes to the segmented passes, 22-bit keys take their runs from bits [6, 22): both end in LDS in the small tile)
    # (refused: all-equal keys -- the ordinary passes are identities, nothing is cheaper -- and 20-bit keys but for ONE key with bit 31
    # set, at an index the sample does not read; the crowded keys are thirty long runs for the segmented passes since round 6)
    missed = _uniform(n, 28) >> np.uint32(12)
    missed[7] |= np.uint32(0x80000000)
    inputs = [(_uniform(n, 21), 1, 1536), (_with_one_run_of(n, 3000, 22), 1, 1536), (_uniform(n, 23) >> np.uint32(10), 1, 1536),
              (np.full(n, 5, dtype=np.uint32), 0, 4608), (crowd, 1, 4608), (missed, 0, 4608), (_with_one_run_of(n, 2000, 24), 1, 1536),
              (_uniform(n, 25), 1, 1536)]
    with torch.cuda.stream(side):
        kt.copy_(torch.from_numpy(inputs[0][0].view(np.int32)))
        vt.copy_(torch.from_numpy(vals.view(np.int32)))
        sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, side.cuda_stream)  # warm-up outside the capture
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            sorter.run_ptr(kt.data_ptr(), vt.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream)
        for keys, accepted, capacity in inputs:
            kt.copy_(torch.from_numpy(keys.view(np.int32)))
            vt.copy_(torch.from_numpy(vals.view(np.int32)))
            graph.replay()
            side.synchronize()
            _check(keys, vals, kt.cpu().numpy().view(np.uint32), vt.cpu().numpy().view(np.uint32))
            fin = sorter.read_finish()
            assert fin["attempted"] == 1 and fin["accepted"] == accepted and fin["capacity"] == capacity, fin


# ---- 64-bit keys: two counting passes on key bits [48, 64), then six rounds inside LDS on the low 48 bits ---------------------

def _run64(G, sorter, keys, vals):
    kb = G.ShaderStorageBuffer(keys)
    if vals is None:
        sorter.sort_keys_ptr(kb.device_ptr(), keys.size, key_bytes=8)
        G.synchronize()
        return kb.get_data(np.uint64), None, sorter.read_finish()
    vb = G.ShaderStorageBuffer(vals)
    sorter(kb, vb, keys.size, 0, key_bytes=8)
    G.synchronize()
    return kb.get_data(np.uint64), vb.get_data(np.uint32), sorter.read_finish()


def _uniform64(n, seed):
    return np.random.default_rng(seed).integers(0, 2**64, n, dtype=np.uint64)


def test_u64_uniform_keys_end_in_lds(G):
    keys, vals = _uniform64(N_SMALL, 31), np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    gk, gv, fin = _run64(G, s, keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["capacity"] == CAP_SMALL
    assert fin["longest_run"] == np.bincount((keys >> np.uint64(48)).astype(np.int64), minlength=65536).max()
    skipped, alone, roles = s.read_plan(8, roles=True)
    assert skipped == [2] * 8  # the eight ordinary passes were the sequence not taken


def test_u64_keys_only_and_duplicates(G):
    rng = np.random.default_rng(32)
    n = (1 << 23) + 99
    keys = (rng.integers(0, 65536, n, dtype=np.uint64) << np.uint64(48)) | (rng.integers(0, 3, n, dtype=np.uint64) * np.uint64(0x010000010001))
    gk, _, fin = _run64(G, _sorter(G, **SMALL), keys, None)
    assert (gk == np.sort(keys, kind="stable")).all() and fin["accepted"] == 1
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run64(G, _sorter(G, **SMALL), keys[:N_SMALL], vals)
    ek, ev = O.stable_sort_pairs(keys[:N_SMALL], vals)
    assert (gk == ek).all() and (gv == ev).all() and fin["accepted"] == 1


def test_u64_small_range(G):
    keys = _uniform64(N_SMALL, 33) >> np.uint64(20)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0, GLU_HIP_SORT_LONG_RUNS=0, **SMALL)
    gk, gv, fin = _run64(G, s, keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    assert fin["attempted"] == 1 and fin["accepted"] == 0
    # (the default: refused too -- six segmented passes over the one long run would move more than the six ordinary passes on the
    # key bytes that vary; the plan's cost rule)
    s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0, **SMALL)
    gk, gv, fin = _run64(G, s, keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    assert fin["attempted"] == 1 and fin["accepted"] == 0


@pytest.mark.parametrize("length,accepted,capacity", [(1536, 1, 1536), (1537, 1, 2560), (4608, 1, 4608), (4609, 0, 4608)])
def test_u64_the_longest_run_decides(G, length, accepted, capacity):
    rng = np.random.default_rng(34)
    keys = rng.integers(0, 2**64, N_SMALL, dtype=np.uint64)
    run = np.uint64(0xBEEF)
    keys[(keys >> np.uint64(48)) == run] ^= np.uint64(1 << 63)
    pos = rng.choice(N_SMALL, size=length, replace=False)
    keys[pos] = (run << np.uint64(48)) | rng.integers(0, 2**48, length, dtype=np.uint64)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    # (the round-4 rule: since round 6 a long run of 64-bit keys goes to segmented passes too, see test_long_runs_of_every_key_kind)
    gk, gv, fin = _run64(G, _sorter(G, GLU_HIP_SORT_LONG_RUNS=0, **SMALL), keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    assert fin["attempted"] == 1 and fin["accepted"] == accepted and fin["longest_run"] == length and fin["capacity"] == capacity


def test_u64_large_sizes_and_the_largest_tile(G):
    """(Round-4 rule, GLU_HIP_SORT_DEVICE_TOP=0: the host takes the runs' key bits from the object's last attempt.)  2^27 full-range keys (tile of 2560), 2^27 keys with the top two bits clear (a quarter of the runs, four times as long:
    the largest tile, 1024 threads x 9) and BASELINE.json's configs[4], 2^28 keys (tile of 4608): sortedness, the value of
    every pair still points at its key, equal keys in input order."""
    import torch

    for n, clear, cap in [((1 << 27) + 77, 0, 2560), ((1 << 27) - 9, 2, 9216), (1 << 28, 0, 4608)]:
        s = _sorter(G, GLU_HIP_SORT_DEVICE_TOP=0)  # (a fresh object: its first sort takes the runs from the whole key's top bits)
        keys = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda:0")
        if clear:
            keys = (keys >> clear) & torch.tensor((1 << (64 - clear)) - 1, dtype=torch.int64, device="cuda:0")
        vals = torch.arange(n, dtype=torch.int32, device="cuda:0")
        k0 = keys.clone()
        torch.cuda.synchronize()
        s.run_ptr(keys.data_ptr(), vals.data_ptr(), n, key_bytes=8)
        G.synchronize()
        fin = s.read_finish()
        assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["capacity"] == cap, (n, fin)
        flipped = keys ^ torch.tensor(-2**63, dtype=torch.int64, device="cuda:0")
        assert bool((flipped[1:] >= flipped[:-1]).all()), "not sorted"
        del flipped
        assert bool((k0[vals.long()] == keys).all()), "a value does not point at its key"
        eq = keys[1:] == keys[:-1]
        assert bool((vals[1:][eq] > vals[:-1][eq]).all()), "equal keys out of input order"
        del keys, vals, k0, eq
        torch.cuda.empty_cache()


# ---- typed keys: the first top-bit pass encodes on load, the in-LDS pass decodes on store -------------------------------------

@pytest.mark.parametrize("name", ["int32", "float32", "int64", "float64"])
@pytest.mark.parametrize("with_vals", [True, False])
def test_typed_keys_end_in_lds(G, name, with_vals):
    """Signed integers and floats in their natural order (negative zero before positive zero, as the order-preserving integer
    code of the bit patterns has it), with values and keys only."""
    dt = np.dtype(name)
    n = (1 << 23) + 1234 if not with_vals else N_SMALL
    rng = np.random.default_rng(41)
    u = rng.integers(0, 2 ** (8 * dt.itemsize), n, dtype=np.uint32 if dt.itemsize == 4 else np.uint64)
    keys = u.view(dt)
    if dt.kind == "f":
        # no NaNs (they have no place in an order): clear the lowest exponent bit of every NaN pattern; a few zeros of both signs
        mant = 23 if dt.itemsize == 4 else 52
        u = np.where(np.isnan(keys), u & ~(u.dtype.type(1) << u.dtype.type(mant)), u)
        keys = u.view(dt).copy()
        keys[::100000] = dt.type(-0.0)
        keys[1::100000] = dt.type(0.0)
    vals = np.arange(n, dtype=np.uint32)
    # (keys-only sorts of 4-byte keys run the line kernel -- and make the attempt -- from 2^25 keys: forced here)
    s = _sorter(G, **SMALL) if with_vals or dt.itemsize == 8 else _sorter(G, GLU_HIP_SORT_LARGE_MIN=1, **SMALL)
    kb = G.ShaderStorageBuffer(keys)
    vb = G.ShaderStorageBuffer(vals) if with_vals else None
    s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr() if with_vals else None, n, name)
    G.synchronize()
    fin = s.read_finish()
    assert fin["attempted"] == 1 and fin["accepted"] == 1, fin
    bits = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
    top = bits.dtype.type(1) << bits.dtype.type(dt.itemsize * 8 - 1)
    code = (bits ^ top) if dt.kind == "i" else np.where(bits & top, ~bits, bits ^ top)
    order = np.argsort(code, kind="stable")
    assert (kb.get_data(dt).view(bits.dtype) == bits[order]).all()
    if with_vals:
        assert (vb.get_data(np.uint32) == vals[order]).all()


def test_typed_keys_of_a_small_range(G):
    n = N_SMALL
    keys = np.random.default_rng(42).integers(-5000, 5000, n, dtype=np.int32)
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    for long_runs, accepted in ((0, 0), (1, 1)):  # (the round-4 rule; the default since round 6: two long runs, the segmented passes')
        s = _sorter(G, GLU_HIP_SORT_LONG_RUNS=long_runs, **SMALL)
        kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
        s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, "int32")
        G.synchronize()
        fin = s.read_finish()
        assert fin["attempted"] == 1 and fin["accepted"] == accepted, fin
        assert (kb.get_data(np.int32) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()


@pytest.mark.parametrize("name", ["int32", "float64"])
def test_typed_keys_in_runs_of_one_are_decoded(G, name):
    """A few negative keys among non-negative ones: every negative key is alone in its run of equal top bits, and a run of one
    needs no sorting -- but it still has to be decoded on its way out (found by tools/fuzz.py: such keys came out encoded)."""
    dt = np.dtype(name)
    rng = np.random.default_rng(43)
    n = N_SMALL
    u = rng.integers(0, 2 ** (8 * dt.itemsize - 1), n, dtype=np.uint32 if dt.itemsize == 4 else np.uint64)  # sign bit clear
    if dt.kind == "f":
        u = np.where(np.isnan(u.view(dt)), u & ~(u.dtype.type(1) << u.dtype.type(52)), u)
    u[rng.choice(n, size=n // 1500, replace=False)] |= u.dtype.type(1) << u.dtype.type(8 * dt.itemsize - 1)
    keys = u.view(dt).copy()
    vals = np.arange(n, dtype=np.uint32)
    s = _sorter(G, **SMALL)
    kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, name)
    G.synchronize()
    assert s.read_finish()["accepted"] == 1
    top = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
    code = (u ^ top) if dt.kind == "i" else np.where(u & top, ~u, u ^ top)
    order = np.argsort(code, kind="stable")
    assert (kb.get_data(dt).view(u.dtype) == u[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()


# ---- 64-bit keys, round 5: the in-LDS pass ranks key bits [32, 48) only (two rounds) and repairs ties on them exactly; runs
# whose keys crowd on those bits fall back to all six rounds inside the same workgroup

def _u64_with_low48(n, seed, low48):
    """uniform top 16 bits (the runs), the low 48 bits from low48(rng, n)"""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 65536, n, dtype=np.uint64) << np.uint64(48)) | (low48(rng, n).astype(np.uint64) & np.uint64((1 << 48) - 1))


@pytest.mark.parametrize("shape", ["ties_on_ranked_bits_reversed_below", "equal_on_16_64_reversed_on_0_16", "all_low_bits_equal",
                                   "pairs_of_ties", "one_tie_group_longer_than_the_repair_bound", "tie_groups_of_equal_keys",
                                   "only_bits_0_32_vary", "walk_back_limit"])
@pytest.mark.parametrize("with_vals", [True, False])
def test_u64_tie_repair_adversarial_shapes(G, shape, with_vals):
    n = N_SMALL
    if shape == "ties_on_ranked_bits_reversed_below":
        # every run: the ranked bits take 8 values, the bits below count down in input order -- every tie group is out of order
        low = lambda rng, n: (rng.integers(0, 8, n).astype(np.uint64) << np.uint64(32)) | (np.uint64(n) - np.arange(n, dtype=np.uint64))
    elif shape == "equal_on_16_64_reversed_on_0_16":
        low = lambda rng, n: np.uint64(0xABCD12340000) | ((np.uint64(n) - np.arange(n, dtype=np.uint64)) & np.uint64(0xFFFF))
    elif shape == "all_low_bits_equal":
        low = lambda rng, n: np.full(n, 0x123456789ABC, dtype=np.uint64)
    elif shape == "pairs_of_ties":
        # neighbours in the input share the ranked bits and come in descending order below them
        def low(rng, n):
            hi = np.repeat(rng.integers(0, 65536, (n + 1) // 2).astype(np.uint64), 2)[:n]
            return (hi << np.uint64(32)) | np.where(np.arange(n) % 2 == 0, np.uint64(7), np.uint64(3))
    elif shape == "one_tie_group_longer_than_the_repair_bound":
        def low(rng, n):
            k = rng.integers(0, 2**48, n, dtype=np.uint64)
            k[: n // 300] = (np.uint64(0x5A5A) << np.uint64(32)) | rng.integers(0, 2**32, n // 300, dtype=np.uint64)  # ~ 14 000 ties, 0.2 per run .. no: spread over all runs
            return k
    elif shape == "tie_groups_of_equal_keys":
        low = lambda rng, n: rng.integers(0, 5, n).astype(np.uint64) * np.uint64(0x111100000001)
    elif shape == "only_bits_0_32_vary":
        low = lambda rng, n: rng.integers(0, 2**32, n, dtype=np.uint64)
    else:  # walk_back_limit: tie groups of 20 in ascending order whose last two are swapped
        def low(rng, n):
            g = np.arange(n, dtype=np.uint64) // np.uint64(20)
            w = np.arange(n, dtype=np.uint64) % np.uint64(20)
            w = np.where(w == 18, np.uint64(19), np.where(w == 19, np.uint64(18), w))
            return ((g % np.uint64(65536)) << np.uint64(32)) | w
    keys = _u64_with_low48(n, 41, low)
    if shape == "one_tie_group_longer_than_the_repair_bound":
        keys[:100] = (np.uint64(0x0042) << np.uint64(48)) | (np.uint64(0x5A5A) << np.uint64(32)) | (np.uint64(1000) - np.arange(100, dtype=np.uint64))
    if shape == "walk_back_limit":
        # groups of 20 must be neighbours inside their run: one run for all
        keys = (keys & np.uint64((1 << 48) - 1)) | (np.repeat(np.arange((n + 1199) // 1200, dtype=np.uint64), 1200)[:n] << np.uint64(48))
    vals = np.arange(n, dtype=np.uint32) if with_vals else None
    gk, gv, fin = _run64(G, _sorter(G, **SMALL), keys, vals)
    assert fin["attempted"] == 1 and fin["accepted"] == 1, fin
    if with_vals:
        ek, ev = O.stable_sort_pairs(keys, vals)
        assert (gk == ek).all() and (gv == ev).all()
    else:
        assert (gk == np.sort(keys, kind="stable")).all()


@pytest.mark.parametrize("rank_bits", [16, 24, 48])
def test_u64_rank_bits_switch(G, rank_bits):
    """GLU_HIP_FINISH_RANK_BITS (tuning): 16 (default) = two rounds + tie repair, 24 = three, 48 = all six rounds as in round 4."""
    keys, vals = _uniform64(N_SMALL, 43), np.arange(N_SMALL, dtype=np.uint32)
    keys[::3] &= np.uint64(0xFFFFFFFF0000FFFF)  # a third of the keys tie on bits [16, 32)
    old = os.environ.get("GLU_HIP_FINISH_RANK_BITS")
    os.environ["GLU_HIP_FINISH_RANK_BITS"] = str(rank_bits)
    try:
        gk, gv, fin = _run64(G, _sorter(G, **SMALL), keys, vals)
    finally:
        if old is None:
            del os.environ["GLU_HIP_FINISH_RANK_BITS"]
        else:
            os.environ["GLU_HIP_FINISH_RANK_BITS"] = old
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all() and fin["accepted"] == 1


# ---- round 5: the runs' key bits are chosen on the device from a sample of the keys: small-range keys end in LDS on an object's
# FIRST sort (radix_sample_top_kernel)

@pytest.mark.parametrize("bits,garbage", [(28, 0), (28, 0xA0000000), (21, 0), (17, 0x00FE0000), (16, 0), (12, 0), (31, 0), (32, 0)])
def test_keys_of_a_smaller_range_end_in_lds_on_the_first_sort(G, bits, garbage):
    """garbage: constant bits above the range (they do not vary: the runs are still taken below them)"""
    rng = np.random.default_rng(bits)
    keys = (rng.integers(0, 1 << bits, N_SMALL, dtype=np.uint64).astype(np.uint32)) | np.uint32(garbage)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    for rep in range(2):
        s = _sorter(G, **SMALL)  # a fresh object every time
        gk, gv, fin = _run(G, s, keys, vals)
        _check(keys, vals, gk, gv)
        assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["top_bit"] == max(bits, 16), (rep, fin)


@pytest.mark.parametrize("bits,top", [(64, 64), (52, 52), (44, 48), (36, 40), (32, 32), (30, 30), (20, 20)])
def test_u64_keys_of_a_smaller_range_on_the_first_sort(G, bits, top):
    """(a digit stays inside one key word: top bits between 33 and 47 move up to 40 or 48)"""
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, 1 << bits if bits < 64 else 2**64, N_SMALL, dtype=np.uint64)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    gk, gv, fin = _run64(G, _sorter(G, **SMALL), keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (gk == ek).all() and (gv == ev).all()
    assert fin["attempted"] == 1 and fin["top_bit"] == top, fin
    assert fin["accepted"] == (1 if N_SMALL / min(2 ** (bits - (top - 16)), 65536) < 1400 else fin["accepted"])


def test_a_key_the_sample_missed_refuses_once_and_is_remembered(G):
    """20-bit keys but for ONE key with bit 31 set, at an index the sample does not read: the exact collection of the leader's
    count kernel sees it, the attempt is refused (the ordinary passes sort), and the next sort's sample starts from that top
    bit; keys that are really small again drop it."""
    rng = np.random.default_rng(99)
    keys = rng.integers(0, 1 << 20, N_SMALL, dtype=np.uint32)
    keys[7] |= np.uint32(0x80000000)
    vals = np.arange(N_SMALL, dtype=np.uint32)
    s = _sorter(G, GLU_HIP_SORT_FINISH_BACKOFF=0, **SMALL)
    gk, gv, fin = _run(G, s, keys, vals)
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 0 and fin["top_bit"] == 20, fin
    gk, gv, fin = _run(G, s, keys, vals)  # (under the whole key's top bits the 20-bit keys crowd into 16 long runs: the segmented passes')
    _check(keys, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["top_bit"] == 32 and s.read_long_runs()["runs"] >= 16, fin
    small = keys & np.uint32(0x000FFFFF)
    gk, gv, fin = _run(G, s, small, vals)  # (still from the remembered top bit: the exact bits say 20)
    _check(small, vals, gk, gv)
    assert fin["top_bit"] == 32
    gk, gv, fin = _run(G, s, small, vals)
    _check(small, vals, gk, gv)
    assert fin["attempted"] == 1 and fin["accepted"] == 1 and fin["top_bit"] == 20, fin
