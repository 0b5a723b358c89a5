"""CPU tests of the host half of the sort that ends in LDS (radix_lds_finish.hpp; glu_radix_sort_plan_finish): from which size
a whole-key sort makes the attempt and which tiles of the in-LDS pass it enqueues.  No device needed: the function is pure."""
import math

import pytest

import glu_hip as G

CAPS = [1536, 2560, 4608, 9216]  # 256 x 6, 256 x 10, 256 x 18, 512 x 18 pairs (64-bit keys: 512 x 9, 1024 x 9)


@pytest.mark.parametrize("key_bytes,first", [(4, 7 << 22), (8, 3 << 21)])
def test_the_attempt_starts_where_it_pays(key_bytes, first):
    assert G.plan_finish(first - 1, key_bytes) == (0, 0)
    assert G.plan_finish(first, key_bytes) == (1536, 4608)


def test_the_tiles_follow_the_mean_run_length():
    """The first tile is the smallest that holds mean + 6 sigma + 8 of 65536 runs of uniformly drawn keys; the next two larger
    ones are enqueued behind it."""
    for lg in (25, 25.5, 26, 26.5, 27, 27.5, 28, 28.5, 29):
        n = int(2 ** lg)
        mean = n / 65536
        need = mean + 6 * math.sqrt(mean) + 8
        want = next(c for c in CAPS if need <= c)
        first, last = G.plan_finish(n)
        assert first == want and last == CAPS[min(CAPS.index(want) + 2, 3)], (lg, first, last)
    assert G.plan_finish(1 << 26) == (1536, 4608)
    assert G.plan_finish(1 << 27) == (2560, 9216)
    assert G.plan_finish(1 << 28) == (4608, 9216)
    assert G.plan_finish(1 << 29) == (9216, 9216)
    assert G.plan_finish(1 << 28, key_bytes=8) == (4608, 9216)


def test_beyond_the_largest_tile_there_is_no_attempt():
    limit = max(n for n in range(1 << 29, (1 << 29) + (1 << 26), 1 << 16) if G.plan_finish(n) != (0, 0))
    mean = (limit + (1 << 16)) / 65536
    assert mean + 6 * math.sqrt(mean) + 8 > 9216  # the next size up would not fit the largest tile with 6 sigma to spare
    assert G.plan_finish((1 << 29) + (1 << 26)) == (0, 0)
    assert G.plan_finish(0xFFFF0000) == (0, 0)


def test_argument_checks():
    with pytest.raises(G.GluError):
        G.plan_finish(1 << 26, key_bytes=2)


def test_the_makefile_tracks_every_header_of_the_library():
    """A header missing from the library's prerequisites lets `make` keep a stale libglu_hip.so after an edit (it happened:
    a fix in radix_lds_finish.hpp was 'tested' against the old binary until the fuzzer failed the same way again)."""
    import glob
    import os

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gl-radix-sort_amd", "csrc")
    text = open(os.path.join(csrc, "Makefile")).read().replace("\\\n", " ")
    listed = set()
    for var in ("KERNEL_HEADERS", "HOST_HEADERS"):
        listed |= set(next(l for l in text.splitlines() if l.startswith(var + " :=")).split()[2:])
    rule = next(l for l in text.splitlines() if l.startswith("$(OBJ)/%.o:"))
    assert "$(KERNEL_HEADERS)" in rule and "$(HOST_HEADERS)" in rule  # (every translation unit depends on every header)
    for header in glob.glob(os.path.join(csrc, "*.hpp")):
        assert os.path.basename(header) in listed, os.path.basename(header)
    units = next(l for l in text.splitlines() if l.startswith("UNITS :=")).split()[2:]
    assert sorted(units) == sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(csrc, "*.hip")))
