"""CPU tests of the host half of the segmented sort (glu_radix_sort_plan_segments: how a segmented pass is cut into
sub-blocks).  No device needed: the function is pure."""
import numpy as np
import pytest

import glu_hip as G


def check_plan(begin, length, seg, nseg, nwg):
    subs, wg_first, seg_first, seg_start = G.plan_segments(begin, length, seg, nseg, nwg)
    begin, length, seg = np.asarray(begin, np.int64), np.asarray(length, np.int64), np.asarray(seg, np.int64)
    total = int(length.sum())
    # every sub-block lies inside exactly one piece; together they cover every piece once, in order
    order = np.argsort(seg, kind="stable")  # segment-major, the caller's order inside a segment
    covered = []
    for b, e in subs.astype(np.int64):
        assert e > b
        covered.append((int(b), int(e)))
    at = 0
    sub_seg = []
    for p in order:
        pos, end = int(begin[p]), int(begin[p] + length[p])
        while pos < end:
            b, e = covered[at]
            assert b == pos and e <= end, (b, e, pos, end)
            sub_seg.append(int(seg[p]))
            pos = e
            at += 1
    assert at == len(covered)
    # segments: contiguous sub-block ranges in ascending segment order; output starts = running sums
    assert seg_first[0] == 0 and seg_first[-1] == len(covered) and (np.diff(seg_first.astype(np.int64)) >= 0).all()
    for g in range(nseg):
        assert all(s == g for s in sub_seg[seg_first[g]:seg_first[g + 1]])
    sizes = np.bincount(seg, weights=length, minlength=nseg).astype(np.int64)
    assert (seg_start.astype(np.int64) == np.concatenate([[0], np.cumsum(sizes)])).all()
    # workgroups: contiguous lists covering all sub-blocks; equal shares of the elements (a sub-block never straddles two shares)
    assert wg_first[0] == 0 and wg_first[-1] == len(covered) and (np.diff(wg_first.astype(np.int64)) >= 0).all()
    share = max(1, -(-total // nwg))
    pos = 0
    for w in range(nwg):
        elems = sum(e - b for b, e in covered[wg_first[w]:wg_first[w + 1]])
        lo = pos
        pos += elems
        assert elems <= share and lo // share == (pos - 1) // share if elems else True, (w, elems, share)
    assert pos == total
    assert len(covered) <= len(begin) + nwg
    return subs, wg_first


@pytest.mark.parametrize("nwg", [1, 7, 248, 256])
@pytest.mark.parametrize("sources,nseg", [(1, 256), (8, 32), (3, 5), (2, 128)])
def test_source_major_shards(sources, nseg, nwg):
    rng = np.random.default_rng(sources * 1000 + nseg + nwg)
    lens = rng.integers(0, 5000, sources * nseg)
    lens[rng.random(lens.size) < 0.1] = 0  # empty pieces
    begin = np.concatenate([[0], np.cumsum(lens)[:-1]])
    seg = np.tile(np.arange(nseg), sources)
    check_plan(begin, lens, seg, nseg, nwg)


def test_pieces_in_any_address_order_and_unused_segments():
    rng = np.random.default_rng(1)
    lens = rng.integers(1, 9000, 40)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    perm = rng.permutation(40)
    seg = rng.integers(0, 6, 40)
    seg[seg == 3] = 4  # segment 3 stays empty
    check_plan(starts[perm], lens[perm], seg, 7, 64)


def test_one_piece_one_segment_splits_evenly():
    subs, wg_first = check_plan([0], [1_000_003], [0], 1, 256)
    assert len(subs) == 256 and (np.diff(wg_first.astype(np.int64)) == 1).all()
    assert int(subs[-1][1]) == 1_000_003


def test_tiny_pieces_pile_up_in_one_workgroup():
    """Hundreds of tiny buckets in a row land in one workgroup's share: what makes glu_dist take the ordinary local sort."""
    lens = np.array([3] * 200 + [1_000_000])
    begin = np.concatenate([[0], np.cumsum(lens)[:-1]])
    subs, wg_first = check_plan(begin, lens, np.arange(201), 201, 256)
    assert int(np.diff(wg_first.astype(np.int64)).max()) >= 200


def test_empty_input_and_argument_checks():
    z64, z32 = np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    subs, wg_first, seg_first, seg_start = G.plan_segments(z64, z64, z32, 3, 4)
    assert len(subs) == 0 and not wg_first.any() and not seg_first.any() and not seg_start.any()
    with pytest.raises(G.GluError):
        G.plan_segments([0], [10], [5], 3, 4)  # segment out of range


def test_an_absurd_segment_count_is_refused_not_indexed():
    """num_segments sizes host vectors (num_segments + 1 entries): 0xFFFFFFFF used to wrap that to zero and index it."""
    import ctypes

    L = G.lib()
    pb = np.array([0], dtype=np.uint64)
    pl = np.array([10], dtype=np.uint64)
    ps = np.array([0], dtype=np.uint32)
    n = ctypes.c_size_t(0)
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    for nseg in (0xFFFFFFFF, (1 << 24) + 1):
        rc = L.glu_radix_sort_plan_segments(pb.ctypes.data_as(u64p), pl.ctypes.data_as(u64p), ps.ctypes.data_as(u32p), 1, nseg, 4,
                                            None, 0, None, None, None, ctypes.byref(n))
        assert rc == G.GLU_ERROR_INVALID_ARGUMENT and b"num_segments" in L.glu_last_error()
    rc = L.glu_radix_sort_plan_segments(pb.ctypes.data_as(u64p), pl.ctypes.data_as(u64p), ps.ctypes.data_as(u32p), 1, 1, 0xFFFFFFFF,
                                        None, 0, None, None, None, ctypes.byref(n))
    assert rc == G.GLU_ERROR_INVALID_ARGUMENT
