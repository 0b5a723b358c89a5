"""CPU tests: the oracle (oracle/glu_oracle.c) against every known-answer vector the reference's tests hold
(tests/golden/reference_vectors.json) and against the reference tests' own assertions on its seeded inputs."""
import numpy as np
import pytest

import oracle as O
from conftest import fnv1a64


def test_minstd_conformance(golden):
    g = golden["reference"]["minstd_rand"]
    r = O.minstd_sample(g["seed"], 10000, 0, 0xFFFFFFFF)
    assert r[:4].tolist() == g["first"]
    assert int(r[9999]) == g["value_10000"]
    # seed 0 = default-constructed engine = seed 1 (test/util/Random.hpp:18-21)
    assert (O.minstd_sample(0, 100, 0, 0xFFFFFFFF) == r[:100]).all()
    # sample_int(min, max) = engine() % (max - min) + min
    assert (O.minstd_sample(1, 100, 5, 15) == (r[:100] % 10) + 5).all()


def test_integer_helpers():
    L = O.lib()
    assert L.glu_oracle_div_ceil(10, 3) == 4 and L.glu_oracle_div_ceil(9, 3) == 3 and L.glu_oracle_div_ceil(1, 1024) == 1
    assert L.glu_oracle_is_power_of_2(0) == 1  # reference quirk, gl_utils.hpp:285-289
    assert L.glu_oracle_is_power_of_2(1024) == 1 and L.glu_oracle_is_power_of_2(1000) == 0
    assert L.glu_oracle_next_power_of_2(1) == 1 and L.glu_oracle_next_power_of_2(3) == 4
    assert L.glu_oracle_next_power_of_2(1024) == 1024 and L.glu_oracle_next_power_of_2(1025) == 2048
    # scratch sizing, RadixSort.hpp:337-353
    assert L.glu_oracle_radix_scratch_buffer_size(1000) == 1024 * 4
    assert L.glu_oracle_radix_block_count_buffer_size(3001) == 16 * 4 * 4
    assert L.glu_oracle_radix_block_count_buffer_size(1 << 28) == 16 * (1 << 18) * 4


def test_scan_simple_known_answer(golden):
    g = golden["reference"]["blelloch_scan_simple"]
    out = O.blelloch_scan_u32(np.array(g["input"], dtype=np.uint32), len(g["input"]))
    assert out.tolist() == g["expected"]


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 65536, 1048576])
def test_scan_reference_sizes(n):
    d = O.minstd_sample(123, n, 0, 100)
    assert (O.blelloch_scan_u32(d, n) == O.exclusive_scan_u32(d, n)).all()
    e = np.zeros(n, dtype=np.uint64)
    e[1:] = np.cumsum(d.astype(np.uint64))[:-1]
    assert (O.exclusive_scan_u32(d, n) == (e & 0xFFFFFFFF)).all()


@pytest.mark.parametrize("parts", [1, 32, 100, 1000])
def test_scan_reference_partitions(parts):
    d = O.minstd_sample(123, 1024 * parts, 0, 100)
    assert (O.blelloch_scan_u32(d, 1024, parts) == O.exclusive_scan_u32(d, 1024, parts)).all()


@pytest.mark.parametrize("count", [1, 2, 4])
def test_scan_degenerate_counts(count):
    # count == 1 is what RadixSort feeds the scan when N <= 1024 (nbp2 = 1): all zeros, including the
    # u_step = 0 downsweep dispatch whose index arithmetic wraps (BlellochScan.hpp:62)
    d = O.minstd_sample(9, count * 16, 0, 1000)
    assert (O.blelloch_scan_u32(d, count, 16) == O.exclusive_scan_u32(d, count, 16)).all()


def test_scan_rejects_what_the_reference_rejects():
    d = np.arange(12, dtype=np.uint32)
    with pytest.raises(ValueError):
        O.blelloch_scan_u32(d, 12)  # not a power of 2, BlellochScan.hpp:134
    with pytest.raises(ValueError):
        O.blelloch_scan_u32(d, 0)
    with pytest.raises(ValueError):
        O.blelloch_scan_u32(d[:8], 8, 0)


def test_reduce_simple_known_answers(golden):
    g = golden["reference"]["reduce_simple_uint"]
    data = np.array(g["input"], dtype=np.uint32)
    for c in g["cases"]:
        assert O.reduce_reference_u32(data, c["count"], c["op"]) == c["expected"]
        assert int(O.reduce_expected(data[:c["count"]], 3, c["op"])[0]) == c["expected"]


def test_reduce_all_known_answers(golden):
    for c in golden["reference"]["reduce_all"]["cases"]:
        got = O.reduce_expected(c["input"], c["data_type"], 0)
        assert np.allclose(np.asarray(got, dtype=np.float64), c["expected"], atol=max(c["abs_tol"], 1e-9), rtol=0)


def test_reduce_reference_sizes(golden):
    g = golden["reference"]["reduce_size_tests"]
    for n in g["fitting"] + [x for x in g["non_fitting"] if x < 400000]:
        d = O.minstd_sample(1, n, 0, 100)
        assert O.reduce_reference_u32(d, n, 0) == int(d.astype(np.uint64).sum() & 0xFFFFFFFF)


def test_reduce_literal_needs_subgroup_32():
    # the shader's stride 32^depth only composes with 32-wide subgroups (Reduce.hpp:26): restated faithfully,
    # a 64-wide subgroup gives a different (wrong) data[0] -- this is why the HIP kernel does not port it
    d = O.minstd_sample(1, 4096, 0, 100)
    assert O.reduce_reference_u32(d, 4096, 0, subgroup_size=32) == int(d.sum())
    assert O.reduce_reference_u32(d, 4096, 0, subgroup_size=64) != int(d.sum())


def test_radix_reference_test_cases(golden):
    """The reference's own assertions (radix_sort_tests.cpp:35-51) on its own inputs, + the committed checksums."""
    sums = {(c["n"], c["max"]): c for c in golden["checksums"]["cases"]}
    for c in golden["reference"]["radix_sort_tests"]["cases"]:
        n = c["n"]
        keys = O.minstd_sample(1, n, c["min"], c["max"])
        vals = np.arange(n, dtype=np.uint32)
        res = O.radix_sort_reference(keys, vals)
        sk, sv = res["result_keys"], res["result_vals"]
        assert res["passes"] == 8 and not res["result_in_scratch"]
        assert (np.diff(sk.astype(np.int64)) >= 0).all()                       # check_sorted
        assert (np.sort(keys) == sk).all()                                     # check_permutation
        ek, ev = O.stable_sort_pairs(keys, vals)
        assert (sk == ek).all() and (sv == ev).all()                           # == stable sort by key
        idx = np.argsort(keys, kind="stable")
        assert (sv == idx.astype(np.uint32)).all()
        s = sums[(n, c["max"])]
        assert fnv1a64(sk) == s["sorted_keys_fnv1a64"] and fnv1a64(sv) == s["sorted_vals_fnv1a64"]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 1023, 1024, 1025, 5000])
def test_radix_literal_equals_stable_sort_full_32_bit(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)  # sets bit 31, which the reference's generator never does
    vals = np.arange(n, dtype=np.uint32)
    res = O.radix_sort_reference(keys, vals)
    ek, ev = O.stable_sort_pairs(keys, vals)
    assert (res["result_keys"] == ek).all() and (res["result_vals"] == ev).all()
    assert res["passes"] == (8 if n > 1 else 0)  # count <= 1 early-out, RadixSort.hpp:278


def test_radix_duplicates_and_constant_keys():
    for keys in (O.minstd_sample(1, 5000, 0, 300), np.zeros(3000, dtype=np.uint32), np.full(3000, 0xFFFFFFFF, dtype=np.uint32)):
        vals = np.arange(keys.size, dtype=np.uint32)
        res = O.radix_sort_reference(keys, vals)
        ek, ev = O.stable_sort_pairs(keys, vals)
        assert (res["result_keys"] == ek).all() and (res["result_vals"] == ev).all()


@pytest.mark.parametrize("steps", [1, 2, 3, 4, 5, 6, 7, 8, 9, 100])
def test_radix_num_steps_quirk(steps):
    """RadixSort.hpp:286-287,331-332: sorts by the low 4*num_steps bits; an odd num_steps leaves the result in the
    scratch buffers and the caller's buffers hold the previous pass."""
    n = 3001
    keys = O.minstd_sample(1, n, 0, 0xFFFFFFFF) ^ np.uint32(0x80000000)
    vals = np.arange(n, dtype=np.uint32)
    res = O.radix_sort_reference(keys, vals, num_steps=steps)
    eff = min(steps, 8)
    assert res["passes"] == eff
    assert res["result_in_scratch"] == (eff % 2 == 1)
    ek, ev = O.stable_sort_pairs(keys, vals, key_bits=4 * eff)
    assert (res["result_keys"] == ek).all() and (res["result_vals"] == ev).all()
    if eff % 2 == 1:
        pk, pv = O.stable_sort_pairs(keys, vals, key_bits=4 * (eff - 1)) if eff > 1 else (keys, vals)
        assert (res["user_keys"] == pk).all() and (res["user_vals"] == pv).all()


def test_radix_trace_tables_match_golden(golden):
    g = golden["checksums"]["trace_n3001"]
    n = 3001
    keys = O.minstd_sample(1, n, 0, 0xFFFFFFFF)
    res = O.radix_sort_reference(keys, np.arange(n, dtype=np.uint32), trace=True)
    assert res["tables"].tolist() == g["tables"]
    assert fnv1a64(res["result_keys"]) == g["sorted_keys_fnv1a64"]
    # pass 0 table: row d = exclusive scan over blocks of the per-block digit counts (RadixSort.hpp:46-47,311)
    nbp2 = 4
    t0 = res["tables"][0].reshape(16, nbp2)
    digits = keys & 0xF
    for d in range(16):
        counts = [int(((digits[b * 1024:(b + 1) * 1024]) == d).sum()) for b in range(3)]
        assert t0[d].tolist() == [0, counts[0], counts[0] + counts[1], counts[0] + counts[1] + counts[2]]


def test_stable_sort_u64():
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**64, 10000, dtype=np.uint64)
    keys[::7] = keys[0]
    vals = np.arange(keys.size, dtype=np.uint32)
    k, v = O.stable_sort_pairs(keys, vals)
    idx = np.argsort(keys, kind="stable")
    assert (k == keys[idx]).all() and (v == idx.astype(np.uint32)).all()


def test_radix_literal_vs_stable_sort_randomized():
    """Many random (size, key range, num_steps): the literal restatement of the shaders equals the stable sort by the
    masked key -- the property every GPU parity test relies on for sizes where the literal form is too slow."""
    rng = np.random.default_rng(2024)
    for _ in range(150):
        n = int(rng.integers(0, 6000))
        hi = int(rng.choice([2, 10, 300, 70000, 2**32]))
        keys = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
        if rng.integers(0, 2):
            keys = keys << np.uint32(rng.integers(0, 24))
        vals = rng.integers(0, 2**32, n, dtype=np.uint32)
        steps = int(rng.integers(0, 10))
        res = O.radix_sort_reference(keys, vals, num_steps=steps)
        eff = 8 if steps == 0 or steps > 8 else steps
        ek, ev = O.stable_sort_pairs(keys, vals, key_bits=4 * eff) if n > 1 else (keys, vals)
        assert (res["result_keys"] == ek).all() and (res["result_vals"] == ev).all(), (n, hi, steps)


def test_oracle_under_address_and_ub_sanitizers():
    """The C oracle (both restatements, scan, reduce, generator) on the reference tests' sizes under ASan + UBSan
    (`make -C oracle sanitize`): no report, and the literal restatement, the LSD checker and plain loops agree.
    Sanitizers run on the CPU build only: there is no GPU ASan on this pool."""
    import os, subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "sanitize"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "oracle selfcheck: 0 failure(s)" in out.stdout
