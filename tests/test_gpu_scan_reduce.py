"""GPU parity tests of glu::BlellochScan and glu::Reduce through the C ABI (reference
test/blelloch_scan_tests.cpp, test/reduce_tests.cpp: same inputs and assertions, + all data types)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return built


def test_scan_simple_known_answer(G, golden):
    g = golden["reference"]["blelloch_scan_simple"]
    b = G.ShaderStorageBuffer(np.array(g["input"], dtype=np.uint32))
    G.BlellochScan(G.DataType_Uint)(b, len(g["input"]))
    assert b.get_data(np.uint32).tolist() == g["expected"]


def test_scan_reference_sizes(G, golden):
    scan = G.BlellochScan(G.DataType_Uint)
    for n in golden["reference"]["blelloch_scan_tests"]["sizes"]:
        d = O.minstd_sample(123, n, 0, 100)
        b = G.ShaderStorageBuffer(d)
        scan(b, n)
        assert (b.get_data(np.uint32) == O.blelloch_scan_u32(d, n)).all(), n  # literal reference algorithm


def test_scan_reference_partitions(G, golden):
    p = golden["reference"]["blelloch_scan_tests"]["partitions"]
    scan = G.BlellochScan(G.DataType_Uint)
    for parts in p["num_partitions"]:
        d = O.minstd_sample(123, p["count"] * parts, 0, 100)
        b = G.ShaderStorageBuffer(d)
        scan(b, p["count"], parts)
        assert (b.get_data(np.uint32) == O.blelloch_scan_u32(d, p["count"], parts)).all()


@pytest.mark.parametrize("count,parts", [(1, 1), (1, 16), (2, 16), (4, 5), (64, 3), (4096, 7), (8192, 3), (1 << 14, 16),
                                         (1 << 22, 2), (1 << 24, 1)])
def test_scan_shapes_u32_wraparound(G, count, parts):
    rng = np.random.default_rng(count + parts)
    d = rng.integers(0, 2**32, count * parts, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    G.BlellochScan(G.DataType_Uint)(b, count, parts)
    assert (b.get_data(np.uint32) == O.exclusive_scan_u32(d, count, parts)).all()


@pytest.mark.parametrize("count", [3, 1000, 4097, 100003, (1 << 20) + 5])
def test_scan_non_power_of_two_through_raw_pointer_entry(G, count):
    rng = np.random.default_rng(count)
    parts = 3
    d = rng.integers(0, 1000, count * parts, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    G.BlellochScan(G.DataType_Uint).run_ptr(b.device_ptr(), count, parts)
    assert (b.get_data(np.uint32) == O.exclusive_scan_u32(d, count, parts)).all()


@pytest.mark.parametrize("chained", ["2", "1", "0"])
def test_scan_chained_and_reduce_then_scan_paths_agree(G, monkeypatch, chained):
    """4-byte types use the single-pass chained scan (decoupled look-back) from 256 chunks of 32768 elements up
    (GLU_HIP_SCAN_CHAINED=2: from 2 chunks up, so the small cases here exercise it too); GLU_HIP_SCAN_CHAINED=0 selects
    the 3-launch reduce-then-scan path that the wider types and the smaller counts always use.  All equal the oracle."""
    monkeypatch.setenv("GLU_HIP_SCAN_CHAINED", chained)
    rng = np.random.default_rng(int(chained))
    for dt, npdt in ((G.DataType_Uint, np.uint32), (G.DataType_Int, np.int32), (G.DataType_Float, np.float32)):
        scan = G.BlellochScan(dt)
        for count, parts in ((1 << 13, 5), (1 << 20, 3), (1 << 25 if npdt != np.float32 else 1 << 22, 1), (32768 * 65 + 7, 2), (32768 * 2 + 1, 7)):
            raw = rng.integers(0, 4, count * parts)
            d = (raw * (0.5 if npdt == np.float32 else 1)).astype(npdt)  # float partial sums stay exact
            b = G.ShaderStorageBuffer(d)
            scan.run_ptr(b.device_ptr(), count, parts)
            got = b.get_data(npdt).reshape(parts, count)
            x = d.reshape(parts, count)
            exp = np.zeros_like(x)
            exp[:, 1:] = np.cumsum(x.astype(np.float64 if npdt == np.float32 else np.int64), axis=1)[:, :-1].astype(npdt)
            assert (got == exp).all(), (dt, count, parts)
        for rep in range(3):  # epochs: the chain words of earlier launches must read as "not ready"
            d = rng.integers(0, 2**32, 1 << 22, dtype=np.uint32)
            b = G.ShaderStorageBuffer(d)
            G.BlellochScan(G.DataType_Uint)(b, 1 << 22)
            assert (b.get_data(np.uint32) == O.exclusive_scan_u32(d, 1 << 22)).all()


def test_scan_argument_checks(G):
    scan = G.BlellochScan(G.DataType_Uint)
    b = G.ShaderStorageBuffer(np.arange(16, dtype=np.uint32))
    for args, msg in (((0, 8), "Invalid buffer"), ((b, 0), "Count must be greater than zero"),
                      ((b, 12), "Count must be a power of 2"), ((b, 8, 0), "Num of partitions must be >= 1")):
        with pytest.raises(G.GluError) as e:  # BlellochScan.hpp:132-135
            scan(*args)
        assert msg in e.value.message
    with pytest.raises(G.GluError):
        scan(b, 16, 2)  # exceeds the buffer


@pytest.mark.parametrize("dt", range(12))
def test_scan_all_data_types(G, dt):
    npdt, comps = O.dtype_info(dt)
    rng = np.random.default_rng(dt)
    count, parts = 1 << 13, 3
    # small integers scaled by a power of two: every partial sum is exact in float32, so == is the right check
    raw = rng.integers(-50, 50, count * parts * comps)
    d = (raw * (0.25 if np.issubdtype(npdt, np.floating) else 1)).astype(npdt)
    b = G.ShaderStorageBuffer(d)
    G.BlellochScan(dt)(b, count, parts)
    got = b.get_data(npdt).reshape(parts, count, comps)
    x = d.reshape(parts, count, comps)
    exp = np.zeros_like(x)
    if np.issubdtype(npdt, np.floating):
        exp[:, 1:] = np.cumsum(x.astype(np.float64), axis=1)[:, :-1].astype(npdt)
    else:
        exp[:, 1:] = (np.cumsum(x.astype(np.int64), axis=1)[:, :-1] & 0xFFFFFFFF).astype(np.uint32).view(npdt).reshape(parts, count - 1, comps) \
            if npdt == np.int32 else (np.cumsum(x.astype(np.uint64), axis=1)[:, :-1] & 0xFFFFFFFF).astype(np.uint32)
    assert (got == exp).all()


@pytest.mark.parametrize("dt", range(12))
def test_scan_many_small_partitions(G, dt):
    """Power-of-two partitions of at most one wave's span (1024 4-byte elements) take the kernel that packs several
    partitions into a workgroup (scan_small_partitions_kernel): partitions inside one lane's vector, inside a wave's
    group, of one group, of several groups; a last workgroup that is not full; an unaligned array."""
    npdt, comps = O.dtype_info(dt)
    rng = np.random.default_rng(100 + dt)
    is_float = np.issubdtype(npdt, np.floating)
    for count in (1, 2, 4, 8, 32, 64, 128, 256, 512, 1024, 2048):
        for parts, offset in ((4097, 0), (37 + (8192 // count), 0), (999 + (4096 // count), 1)):
            raw = rng.integers(-50, 50, (count * parts + offset) * comps)
            d = (raw * (0.25 if is_float else 1)).astype(npdt)  # every partial sum exact in float32
            b = G.ShaderStorageBuffer(d)
            esize = d.itemsize * comps
            G.BlellochScan(dt).run_ptr(b.device_ptr() + offset * esize, count, parts)
            got = b.get_data(npdt)
            assert (got[:offset * comps] == d[:offset * comps]).all()
            got = got[offset * comps:].reshape(parts, count, comps)
            x = d[offset * comps:].reshape(parts, count, comps)
            exp = np.zeros_like(x)
            if count > 1:
                c = np.cumsum(x.astype(np.float64 if is_float else np.int64), axis=1)[:, :-1]
                exp[:, 1:] = c.astype(npdt) if is_float else (c & 0xFFFFFFFF).astype(np.uint32).view(npdt).reshape(parts, count - 1, comps)
            assert (got == exp).all(), (dt, count, parts, offset)


def test_reduce_simple_known_answers(G, golden):
    g = golden["reference"]["reduce_simple_uint"]
    data = np.array(g["input"], dtype=np.uint32)
    for c in g["cases"]:
        b = G.ShaderStorageBuffer(data)
        G.Reduce(G.DataType_Uint, c["op"])(b, c["count"])
        out = b.get_data(np.uint32)
        assert int(out[0]) == c["expected"]
        assert (out[1:] == data[1:]).all()  # only data[0] is written


def test_reduce_all_known_answers(G, golden):
    for c in golden["reference"]["reduce_all"]["cases"]:
        npdt, comps = O.dtype_info(c["data_type"])
        d = np.array(c["input"], dtype=npdt)
        b = G.ShaderStorageBuffer(d)
        G.Reduce(c["data_type"], G.ReduceOperator_Sum)(b, d.size // comps)
        got = b.get_data(npdt)[:comps].astype(np.float64)
        assert np.allclose(got, c["expected"], rtol=0, atol=max(c["abs_tol"], 1e-9)), c  # reference: WithinAbs 0.1


def test_reduce_reference_sizes(G, golden):
    g = golden["reference"]["reduce_size_tests"]
    red = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)
    for n in g["fitting"] + g["non_fitting"]:
        d = O.minstd_sample(1, n, 0, 100)
        b = G.ShaderStorageBuffer(d)
        red(b, n)
        assert int(b.get_data(np.uint32)[0]) == int(d.astype(np.uint64).sum() & 0xFFFFFFFF), n


@pytest.mark.parametrize("dt", range(12))
@pytest.mark.parametrize("op", range(4))
def test_reduce_every_type_and_operator(G, dt, op):
    npdt, comps = O.dtype_info(dt)
    rng = np.random.default_rng(dt * 4 + op)
    for n in (1, 7, 64, 1000, 262147, 3000001):
        if op == 1:  # products: mostly ones so nothing overflows / underflows
            d = np.ones(n * comps, dtype=npdt)
            d[rng.integers(0, n * comps, 4)] = 2
            if np.issubdtype(npdt, np.signedinteger) or np.issubdtype(npdt, np.floating):
                d[rng.integers(0, n * comps, 3)] = -1
        elif np.issubdtype(npdt, np.floating):
            d = (rng.integers(-4000, 4000, n * comps) * 0.125).astype(npdt)  # exact sums
        elif npdt == np.int32:
            d = rng.integers(-2**31, 2**31, n * comps).astype(npdt)
        else:
            d = rng.integers(0, 2**32, n * comps, dtype=np.uint32)
        b = G.ShaderStorageBuffer(d)
        G.Reduce(dt, op)(b, n)
        got = b.get_data(npdt)[:comps]
        exp = np.asarray(O.reduce_expected(d, dt, op))
        if np.issubdtype(npdt, np.floating):
            assert np.allclose(got.astype(np.float64), exp.astype(np.float64), rtol=1e-6, atol=1e-6), (n, got, exp)
        else:
            assert (got == exp.astype(npdt)).all(), (n, got, exp)


def test_reduce_argument_checks(G):
    b = G.ShaderStorageBuffer(np.arange(16, dtype=np.uint32))
    red = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)
    with pytest.raises(G.GluError) as e:
        red(0, 4)
    assert "Invalid buffer" in e.value.message  # Reduce.hpp:113
    with pytest.raises(G.GluError) as e:
        red(b, 0)
    assert "Count must be greater than zero" in e.value.message  # Reduce.hpp:114
    with pytest.raises(G.GluError):
        G.Reduce(G.DataType_Uint, 7)  # Reduce.hpp:94-97
    with pytest.raises(G.GluError):
        G.Reduce(99, 0)  # data_types.hpp:41


def test_full_size_scan_and_reduce_2_28(G):
    """README's largest benchmark point for scan / reduce (2^28 uint32)."""
    n = 1 << 28
    rng = np.random.default_rng(2)
    d = rng.integers(0, 2**32, n, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)(b, n)
    assert int(b.get_data(np.uint32)[0]) == int(d.sum(dtype=np.uint64) & 0xFFFFFFFF)
    b.write_data(d)
    G.BlellochScan(G.DataType_Uint)(b, n)
    got = b.get_data(np.uint32)
    exp = np.cumsum(d, dtype=np.uint32)  # wraps mod 2^32
    assert got[0] == 0 and (got[1:] == exp[:-1]).all()


def test_buffer_object_semantics(G):
    """ShaderStorageBuffer behaviour the tests rely on (gl_utils.hpp:146-246)."""
    b = G.ShaderStorageBuffer(size=64)
    assert b.handle() != 0 and b.size() == 64
    b.clear(0xDEADBEEF)
    assert (b.get_data(np.uint32) == 0xDEADBEEF).all()
    b.write_data(np.arange(8, dtype=np.uint32))
    h0 = b.handle()
    b.resize(128, keep_data=True)
    assert b.size() == 128 and b.handle() != h0
    assert (b.get_data(np.uint32)[:8] == np.arange(8)).all()
    with pytest.raises(G.GluError):
        b.get_data(np.dtype([("a", np.uint8, 3)]))  # size not a multiple of sizeof(T), gl_utils.hpp:232
    ns = G.measure_elapsed_time(lambda: b.clear(1))
    assert 0 < ns < 10**9
    empty = G.ShaderStorageBuffer()
    assert empty.handle() == 0 and empty.size() == 0


@pytest.mark.parametrize("count,parts", [(1 << 32, 1), (1 << 31, 3), ((1 << 32) + 12345, 1)])
def test_scan_beyond_32_bit_counts_on_device(G, count, parts):
    """Element indices past 2^32 (16 - 24 GiB of uint32): checked on the device against torch.cumsum in int64, chunk by
    chunk with the carried total, modulo 2^32 as the operator wraps."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs 120 GiB of free HBM")
    n = count * parts
    gen = torch.Generator(device="cuda").manual_seed(parts)
    data = torch.empty(n, dtype=torch.int32, device="cuda")
    step = 1 << 28
    for lo in range(0, n, step):
        m = min(step, n - lo)
        data[lo:lo + m] = torch.randint(0, 1 << 20, (m,), generator=gen, device="cuda", dtype=torch.int32)
    orig = data.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        G.BlellochScan(G.DataType_Uint).run_ptr(data.data_ptr(), count, parts, side.cuda_stream)
    side.synchronize()
    for p in range(parts):
        carry = 0
        for lo in range(0, count, step):
            m = min(step, count - lo)
            x = orig[p * count + lo:p * count + lo + m].to(torch.int64)
            inc = torch.cumsum(x, 0)
            exp = (inc - x + carry) & 0xFFFFFFFF
            got = data[p * count + lo:p * count + lo + m].to(torch.int64) & 0xFFFFFFFF
            assert bool((got == exp).all()), (p, lo)
            carry += int(inc[-1])
            del x, inc, exp, got
    del data, orig
    torch.cuda.empty_cache()


@pytest.mark.parametrize("count", [(1 << 33) + 7])
def test_reduce_beyond_32_bit_counts_on_device(G, count):
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 80 * (1 << 30):
        pytest.skip("needs 80 GiB of free HBM")
    gen = torch.Generator(device="cuda").manual_seed(3)
    data = torch.empty(count, dtype=torch.int32, device="cuda")
    step = 1 << 28
    for lo in range(0, count, step):
        m = min(step, count - lo)
        data[lo:lo + m] = torch.randint(-(1 << 20), 1 << 20, (m,), generator=gen, device="cuda", dtype=torch.int32)
    data[count - 1] = (1 << 20) + 5  # the maximum sits in the last element, past index 2^33
    total = sum(int(data[lo:min(count, lo + step)].sum(dtype=torch.int64)) for lo in range(0, count, step))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for op, expect in ((G.ReduceOperator_Max, (1 << 20) + 5), (G.ReduceOperator_Sum, None)):
        work = data.clone()
        side.wait_stream(torch.cuda.current_stream())  # the clone runs on torch's current stream
        with torch.cuda.stream(side):
            G.Reduce(G.DataType_Int, op).run_ptr(work.data_ptr(), count, side.cuda_stream)
        side.synchronize()
        got = int(work[0].item())
        if expect is None:
            expect = ((total + (1 << 31)) % (1 << 32)) - (1 << 31)  # int32 wrap
        assert got == expect, (op, got, expect)
        del work
    del data
    torch.cuda.empty_cache()


def test_scan_captured_into_a_graph_replays_correctly(G):
    """The single-pass chained scan keeps a host-side epoch and is not capturable; a scan issued under stream capture takes
    the reduce-then-scan path instead (after glu_scan_prepare nothing is allocated), so replays give fresh results even
    at a size where the chained kernel would run (2^24 elements)."""
    import torch

    n = 1 << 24
    scan = G.BlellochScan(G.DataType_Uint)
    scan.prepare(n)
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    rng = np.random.default_rng(5)
    with torch.cuda.stream(side):
        data = rng.integers(0, 1000, n, dtype=np.uint32)
        t.copy_(torch.from_numpy(data.view(np.int32)))
        scan.run_ptr(t.data_ptr(), n, 1, side.cuda_stream)  # warm-up outside the capture (chained path)
        side.synchronize()
        expect = np.concatenate([np.zeros(1, np.uint64), np.cumsum(data[:-1], dtype=np.uint64)]).astype(np.uint32)
        assert (t.cpu().numpy().view(np.uint32) == expect).all()
        with torch.cuda.graph(graph, stream=side):
            scan.run_ptr(t.data_ptr(), n, 1, torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            data = rng.integers(0, 1000 + rep, n, dtype=np.uint32)
            t.copy_(torch.from_numpy(data.view(np.int32)))
            graph.replay()
            side.synchronize()
            expect = np.concatenate([np.zeros(1, np.uint64), np.cumsum(data[:-1], dtype=np.uint64)]).astype(np.uint32)
            assert (t.cpu().numpy().view(np.uint32) == expect).all()
