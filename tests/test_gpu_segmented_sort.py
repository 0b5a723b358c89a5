"""GPU parity tests of the segmented stable sort (glu_radix_sort_run_segments_ptr): the reference's stable counting pass
(glu/RadixSort.hpp:142-182) applied per segment, which is the local sort of the sharded sort.  Expected results come
from the oracle: every segment's pieces laid end to end, stably sorted by the low key bits."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return built


def expected(keys, vals, begin, length, seg, nseg, key_bits):
    ek, ev = [], []
    for g in range(nseg):
        idx = [i for i in range(len(seg)) if seg[i] == g]
        k = np.concatenate([keys[begin[i]:begin[i] + length[i]] for i in idx] + [np.zeros(0, np.uint32)])
        v = np.concatenate([vals[begin[i]:begin[i] + length[i]] for i in idx] + [np.zeros(0, np.uint32)])
        if key_bits and k.size:
            k, v = O.stable_sort_pairs(k, v, key_bits)
        ek.append(k)
        ev.append(v)
    return np.concatenate(ek), np.concatenate(ev)


def run(G, keys, vals, begin, length, seg, nseg, key_bits, sorter=None, prepare=True):
    import torch

    n = keys.size
    kin = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    vin = torch.from_numpy(vals.view(np.int32).copy()).cuda()
    kout = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    vout = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    sorter = sorter or G.RadixSort()
    if prepare:
        sorter.prepare_internal_buffers(n)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        sorter.run_segments_ptr(kin.data_ptr(), vin.data_ptr(), kout.data_ptr(), vout.data_ptr(), n, begin, length, seg, nseg,
                                key_bits, st.cuda_stream)
        gk, gv = kout[:n].cpu().numpy().view(np.uint32), vout[:n].cpu().numpy().view(np.uint32)
    return gk, gv


def source_major_pieces(rng, n, sources, nseg, skew=None):
    """The layout a rank receives: one message per source, each grouped by segment (bucket).  Returns piece arrays in
    (source, segment) order -- the order of a segment's pieces is the source order -- plus the keys' segment per element."""
    weights = np.ones(nseg) if skew is None else np.asarray(skew, dtype=np.float64)
    weights = weights / weights.sum()
    per_source = np.diff(np.linspace(0, n, sources + 1).astype(np.int64))
    begin, length, seg = [], [], []
    at = 0
    for s in range(sources):
        counts = rng.multinomial(per_source[s], weights)
        for g in range(nseg):
            begin.append(at)
            length.append(int(counts[g]))
            seg.append(g)
            at += int(counts[g])
    assert at == n
    return np.array(begin, np.uint64), np.array(length, np.uint64), np.array(seg, np.uint32)


@pytest.mark.parametrize("key_bits", [24, 8, 16, 32, 0])
@pytest.mark.parametrize("n,sources,nseg", [(300_000, 8, 32), (1_500_003, 3, 5), (4_200_000, 2, 128)])
def test_segmented_sort_matches_oracle(G, n, sources, nseg, key_bits):
    rng = np.random.default_rng(n + key_bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    keys[::5] &= np.uint32(0xFF00FFFF)  # duplicate-heavy middle byte
    vals = np.arange(n, dtype=np.uint32)
    begin, length, seg = source_major_pieces(rng, n, sources, nseg)
    gk, gv = run(G, keys, vals, begin, length, seg, nseg, key_bits)
    ek, ev = expected(keys, vals, begin, length, seg, nseg, key_bits)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("case", ["one_hot_segment", "empty_segments", "tiny_pieces", "single_segment", "few_distinct_keys",
                                  "pieces_out_of_address_order"])
def test_segmented_sort_structured_cases(G, case):
    rng = np.random.default_rng(7)
    n = 1_000_003
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    if case == "one_hot_segment":
        begin, length, seg = source_major_pieces(rng, n, 8, 32, skew=[1] * 31 + [400])
        nseg = 32
    elif case == "empty_segments":
        begin, length, seg = source_major_pieces(rng, n, 4, 64, skew=[0, 1, 0, 0, 3] + [0] * 58 + [1])
        nseg = 64
    elif case == "tiny_pieces":
        begin, length, seg = source_major_pieces(rng, n, 8, 256, skew=[1e-5] * 255 + [1])
        nseg = 256
    elif case == "single_segment":
        begin, length, seg = np.array([0], np.uint64), np.array([n], np.uint64), np.array([0], np.uint32)
        nseg = 1
    elif case == "few_distinct_keys":
        keys = rng.integers(0, 3, n, dtype=np.uint32) * np.uint32(0x00010101)
        begin, length, seg = source_major_pieces(rng, n, 8, 32)
        nseg = 32
    else:
        # the caller's piece order, not the address order, is the order of a segment's elements
        cuts = np.sort(rng.choice(np.arange(1, n), 19, replace=False))
        b = np.concatenate([[0], cuts])
        l = np.diff(np.concatenate([b, [n]]))
        perm = rng.permutation(20)
        begin, length, seg = b[perm].astype(np.uint64), l[perm].astype(np.uint64), rng.integers(0, 4, 20).astype(np.uint32)
        nseg = 4
    gk, gv = run(G, keys, vals, begin, length, seg, nseg, 24)
    ek, ev = expected(keys, vals, begin, length, seg, nseg, 24)
    assert (gk == ek).all() and (gv == ev).all()


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 40_000, 65_535, 65_536, 65_537])
def test_segmented_sort_small_counts(G, n):
    """Below 2^16 elements the call gathers the pieces and sorts every segment with the ordinary sort."""
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    begin, length, seg = source_major_pieces(rng, n, 3, 7)
    gk, gv = run(G, keys, vals, begin, length, seg, 7, 24)
    ek, ev = expected(keys, vals, begin, length, seg, 7, 24)
    assert (gk == ek).all() and (gv == ev).all()


def test_segmented_sort_reuses_one_object_back_to_back(G):
    """Consecutive calls on one object and one stream share its descriptor scratch: the pinned images are a ring."""
    import torch

    sorter = G.RadixSort()
    rng = np.random.default_rng(3)
    n = 400_000
    sorter.prepare_internal_buffers(n)
    st = torch.cuda.Stream()
    outs, exps = [], []
    with torch.cuda.stream(st):
        for it in range(7):
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
            vals = np.arange(n, dtype=np.uint32)
            begin, length, seg = source_major_pieces(rng, n, 1 + it, 3 + 5 * it)
            kin = torch.from_numpy(keys.view(np.int32)).cuda()
            vin = torch.from_numpy(vals.view(np.int32)).cuda()
            kout, vout = torch.empty_like(kin), torch.empty_like(vin)
            sorter.run_segments_ptr(kin.data_ptr(), vin.data_ptr(), kout.data_ptr(), vout.data_ptr(), n, begin, length, seg,
                                    3 + 5 * it, 24, st.cuda_stream)
            outs.append((kin, vin, kout, vout))
            exps.append(expected(keys, vals, begin, length, seg, 3 + 5 * it, 24))
        st.synchronize()
    for (_, _, kout, vout), (ek, ev) in zip(outs, exps):
        assert (kout.cpu().numpy().view(np.uint32) == ek).all() and (vout.cpu().numpy().view(np.uint32) == ev).all()


def test_segmented_sort_argument_checks(G):
    import torch

    s = G.RadixSort()
    k = torch.zeros(100, dtype=torch.int32, device="cuda")
    v = torch.zeros(100, dtype=torch.int32, device="cuda")
    k2, v2 = torch.zeros_like(k), torch.zeros_like(v)
    one = (np.array([0], np.uint64), np.array([100], np.uint64), np.array([0], np.uint32))
    with pytest.raises(G.GluError):  # in == out
        s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k.data_ptr(), v2.data_ptr(), 100, *one, 1, 24)
    with pytest.raises(G.GluError):  # pieces do not add up
        s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(), 100, one[0], np.array([99], np.uint64), one[2], 1, 24)
    with pytest.raises(G.GluError):  # segment out of range
        s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(), 100, one[0], one[1], np.array([1], np.uint32), 1, 24)
    with pytest.raises(G.GluError):  # key_bits not a multiple of 8
        s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(), 100, *one, 1, 12)
    s.run_segments_ptr(0, 0, 0, 0, 0, np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32), 0, 24)  # empty: fine
    # under stream capture the call is refused (its descriptors are staged per call): the graph stays empty
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    s.prepare_internal_buffers(100)
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            with pytest.raises(G.GluError) as e:
                s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(), 100, *one, 1, 24,
                                   stream=torch.cuda.current_stream().cuda_stream)
            assert "captured" in e.value.message
    s.run_segments_ptr(k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(), 100, *one, 1, 24)  # and works afterwards
    torch.cuda.synchronize()


# ---- a segmented sort that ends in LDS (round 5): one counting pass on the top digit of the bits (into the sorter's scratch) +
# one in-LDS pass over the runs (segment, top digit) into the output; runs longer than a tile are walked in pieces or split over
# several workgroups; the device decides by the longest run, the ordinary passes run behind it otherwise

def _run_with_report(G, keys, vals, begin, length, seg, nseg, key_bits, env=None):
    import os

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        sorter = G.RadixSort()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    gk, gv = run(G, keys, vals, begin, length, seg, nseg, key_bits, sorter=sorter)
    return gk, gv, sorter.read_seg_finish()


@pytest.mark.parametrize("key_bits", [16, 24, 32])
@pytest.mark.parametrize("n,sources,nseg", [(700_001, 8, 32), (3_000_000, 2, 128), (2_500_000, 3, 1)])
def test_segmented_sort_ends_in_lds(G, n, sources, nseg, key_bits):
    """Uniform keys: the device accepts, and the result is the oracle's.  One segment of 2.5 M pairs has runs of 9800."""
    rng = np.random.default_rng(n * 3 + key_bits)
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    begin, length, seg = source_major_pieces(rng, n, sources, nseg)
    gk, gv, rep = _run_with_report(G, keys, vals, begin, length, seg, nseg, key_bits)
    ek, ev = expected(keys, vals, begin, length, seg, nseg, key_bits)
    assert (gk == ek).all() and (gv == ev).all()
    if nseg == 1:
        # runs of 9800 pairs outgrow the largest tile that shares a CU: by default no attempt (the ordinary passes); the two ways
        # to take such runs are switches -- one workgroup per CU with a tile of 17408, or every run split over four workgroups
        assert rep["attempted"] == 0, rep
        for env, tile, split in (({"GLU_HIP_SEG_MAX_GEO": "5"}, 17408, 1), ({"GLU_HIP_SEG_SPLIT_MAX": "3"}, 4608, 4)):
            gk, gv, rep = _run_with_report(G, keys, vals, begin, length, seg, nseg, key_bits, env=env)
            assert (gk == ek).all() and (gv == ev).all()
            assert rep["accepted"] == 1 and rep["tile"] == tile and rep["split"] == split, rep
        return
    assert rep["attempted"] == 1 and rep["accepted"] == 1 and rep["runs"] == nseg * 256, rep
    assert 0 < rep["longest_run"] <= rep["tile"] and rep["split"] == 1, rep
    # and the same input by the ordinary passes (the attempt switched off) gives the same arrays
    pk, pv, rep0 = _run_with_report(G, keys, vals, begin, length, seg, nseg, key_bits, env={"GLU_HIP_SEG_LDS_FINISH": "0"})
    assert rep0["attempted"] == 0
    assert (pk == ek).all() and (pv == ev).all()


@pytest.mark.parametrize("split", [1, 2, 3])
@pytest.mark.parametrize("shape", ["uniform", "skewed_parts", "few_values"])
def test_segmented_sort_runs_split_over_workgroups(G, split, shape):
    """Every run split over 2, 4, 8 workgroups by ranges of the low bits (what the sharded sort does at eight ranks), forced on
    runs that would fit one tile; parts that outgrow their tile (skew) are walked in pieces."""
    rng = np.random.default_rng(split * 7 + len(shape))
    n, nseg, sources = 1_200_000, 4, 5
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    if shape == "skewed_parts":
        keys[rng.random(n) < 0.8] &= np.uint32(0xFFFF07FF)  # 80 % of every run in the lowest of 32 ranges of the low 16 bits
    elif shape == "few_values":
        keys = (keys & np.uint32(0xFFFF0000)) | (rng.integers(0, 3, n, dtype=np.uint32) * np.uint32(0x5555))
    vals = np.arange(n, dtype=np.uint32)
    begin, length, seg = source_major_pieces(rng, n, sources, nseg)
    gk, gv, rep = _run_with_report(G, keys, vals, begin, length, seg, nseg, 24, env={"GLU_HIP_SEG_SPLIT_MIN": str(split), "GLU_HIP_SEG_SPLIT_MAX": "3"})
    ek, ev = expected(keys, vals, begin, length, seg, nseg, 24)
    assert (gk == ek).all() and (gv == ev).all()
    assert rep["accepted"] == 1 and rep["split"] == 1 << split, rep


@pytest.mark.parametrize("shape", ["one_long_run", "run_of_exactly_a_tile", "one_pair_more_than_a_tile", "one_key_value_longer_than_a_tile",
                                   "long_run_crowded_low_bits", "run_at_the_gate", "run_beyond_the_gate", "all_keys_equal",
                                   "empty_segments", "long_run_in_last_segment"])
def test_segmented_sort_lds_ending_long_runs(G, shape):
    """1 M pairs in 16 segments: the uniform-keys tile holds 1536 pairs.  Runs longer than that are walked in pieces that fit
    (radix_finish_ranges_kernel), whatever their keys; a run beyond 32 tiles sends the sort to the ordinary passes."""
    rng = np.random.default_rng(11)
    n, nseg, sources = 1_048_576, 16, 4
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    skew = [1, 0, 0, 2, 0, 1, 1, 0, 1, 1, 0, 0, 0, 1, 0, 0] if shape == "empty_segments" else None
    begin, length, seg = source_major_pieces(rng, n, sources, nseg, skew=skew)

    def seg_indices(g):  # the elements of segment g, through their indices in the input
        return np.concatenate([np.arange(begin[i], begin[i] + length[i]) for i in range(len(seg)) if seg[i] == g]).astype(np.int64)

    def fill_run(g, m, low=None):
        """exactly m pairs of segment g get the top digit 0xA5 (low: their low 16 bits)"""
        idx = seg_indices(g)
        in_run = ((keys[idx] >> 16) & 0xFF) == 0xA5
        keys[idx[in_run]] ^= np.uint32(0x00010000)
        pick = idx[rng.choice(idx.size, m, replace=False)]
        keys[pick] = (keys[pick] & np.uint32(0xFF00FFFF)) | np.uint32(0x00A50000)
        if low is not None:
            keys[pick] = (keys[pick] & np.uint32(0xFFFF0000)) | low(m).astype(np.uint32)
        return m

    want_accept, longest = True, None
    if shape == "one_long_run":
        longest = fill_run(5, 6000)
    elif shape == "long_run_in_last_segment":
        longest = fill_run(nseg - 1, 7001)
    elif shape == "run_of_exactly_a_tile":
        longest = fill_run(5, 1536)
    elif shape == "one_pair_more_than_a_tile":
        longest = fill_run(5, 1537)
    elif shape == "one_key_value_longer_than_a_tile":
        longest = fill_run(5, 5000, low=lambda m: np.where(np.arange(m) % 10 == 0, rng.integers(0, 65536, m), 0x1234))
    elif shape == "long_run_crowded_low_bits":
        longest = fill_run(5, 9000, low=lambda m: np.where(rng.random(m) < 0.9, rng.integers(0, 40, m), rng.integers(0, 65536, m)))
    elif shape == "run_at_the_gate":
        longest = fill_run(5, 1536 * 32)
    elif shape == "run_beyond_the_gate":
        longest = fill_run(5, 1536 * 32 + 1)
        want_accept = False
    elif shape == "all_keys_equal":
        keys[:] = 0x12345678
        want_accept = False
    gk, gv, rep = _run_with_report(G, keys, vals, begin, length, seg, nseg, 24)
    ek, ev = expected(keys, vals, begin, length, seg, nseg, 24)
    assert (gk == ek).all() and (gv == ev).all()
    assert rep["attempted"] == 1 and rep["tile"] == 1536 and rep["split"] == 1 and rep["capacity"] == 1536 * 32, rep
    assert rep["accepted"] == (1 if want_accept else 0), rep
    if longest is not None:
        assert rep["longest_run"] == longest, rep


def test_segmented_sort_lds_ending_back_to_back_with_changing_outcomes(G):
    """One object, one stream, no host synchronisation between the sorts: accepted, refused, accepted -- the gate word and the run
    starts of a sort are rewritten by the next one in stream order."""
    import torch

    sorter = G.RadixSort()
    rng = np.random.default_rng(5)
    n, nseg = 600_000, 2
    sorter.prepare_internal_buffers(n)
    st = torch.cuda.Stream()
    outs, exps = [], []
    with torch.cuda.stream(st):
        for it in range(6):
            keys = rng.integers(0, 2**32, n, dtype=np.uint32)
            if it % 2:
                keys[: n // 2] &= np.uint32(0xFF00FFFF)  # half of the pairs in the two runs (g, 0): 150 000 each, beyond 32 tiles of 1536
            vals = np.arange(n, dtype=np.uint32)
            begin, length, seg = source_major_pieces(rng, n, 2 + it, nseg)
            kin = torch.from_numpy(keys.view(np.int32)).cuda()
            vin = torch.from_numpy(vals.view(np.int32)).cuda()
            kout, vout = torch.empty_like(kin), torch.empty_like(vin)
            sorter.run_segments_ptr(kin.data_ptr(), vin.data_ptr(), kout.data_ptr(), vout.data_ptr(), n, begin, length, seg, nseg, 24,
                                    st.cuda_stream)
            outs.append((kin, vin, kout, vout))
            exps.append(expected(keys, vals, begin, length, seg, nseg, 24))
        st.synchronize()
    for (_, _, kout, vout), (ek, ev) in zip(outs, exps):
        assert (kout.cpu().numpy().view(np.uint32) == ek).all() and (vout.cpu().numpy().view(np.uint32) == ev).all()
    rep = sorter.read_seg_finish()
    assert rep["attempted"] == 1 and rep["accepted"] == 0, rep  # (the last one was a refused one)
