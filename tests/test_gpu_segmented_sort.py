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
