"""Scratch correctness + timing probe (run on the GPU box): python tools/quick_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import glu_hip as G
import oracle as O

print(G.device_info(), flush=True)
rng = np.random.default_rng(1)
ok_all = True
for bits in (4, 8):
    rs = G.RadixSort(digit_bits=bits)
    for n in [2, 3, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 10993, 47487, 100000, 1 << 20, (1 << 22) + 12345]:
        for kind in ("rand", "dup", "zero"):
            if kind == "rand":
                k = rng.integers(0, 2**32, n, dtype=np.uint32)
            elif kind == "dup":
                k = rng.integers(0, 10, n, dtype=np.uint32)
            else:
                k = np.zeros(n, dtype=np.uint32)
            v = np.arange(n, dtype=np.uint32)
            kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(v)
            rs(kb, vb, n)
            gk, gv = kb.get_data(np.uint32), vb.get_data(np.uint32)
            ek, ev = O.stable_sort_pairs(k, v)
            ok = (gk == ek).all() and (gv == ev).all()
            ok_all &= ok
            if not ok:
                bad = np.nonzero((gk != ek) | (gv != ev))[0]
                print("FAIL bits", bits, "n", n, kind, "first bad", bad[:5], gk[bad[:5]], ek[bad[:5]], flush=True)
    for steps in range(1, 9):
        n = 50000
        k = rng.integers(0, 2**32, n, dtype=np.uint32); v = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(v)
        rs(kb, vb, n, steps)
        ek, ev = O.stable_sort_pairs(k, v, key_bits=4 * steps)
        ok = (kb.get_data(np.uint32) == ek).all() and (vb.get_data(np.uint32) == ev).all()
        ok_all &= ok
        if not ok: print("FAIL steps", bits, steps)
    # u64
    for n in [5, 4097, 300000]:
        k = rng.integers(0, 2**64, n, dtype=np.uint64); v = np.arange(n, dtype=np.uint32)
        kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(v)
        rs(kb, vb, n, 0, key_bytes=8)
        ek, ev = O.stable_sort_pairs(k, v)
        ok = (kb.get_data(np.uint64) == ek).all() and (vb.get_data(np.uint32) == ev).all()
        ok_all &= ok
        if not ok: print("FAIL u64", bits, n)
    print("sort bits", bits, "ok so far:", ok_all, flush=True)

# scan
for dt, npdt in [(G.DataType_Uint, np.uint32), (G.DataType_Int, np.int32), (G.DataType_Float, np.float32), (G.DataType_Double, np.float64)]:
    sc = G.BlellochScan(dt)
    for n, parts in [(1, 1), (2, 3), (8, 1), (1024, 1), (1024, 100), (4096, 1), (8192, 3), (1 << 20, 1), (1 << 24, 1), (2048, 16)]:
        d = rng.integers(0, 100, n * parts).astype(npdt)
        b = G.ShaderStorageBuffer(d)
        sc(b, n, parts)
        g = b.get_data(npdt).reshape(parts, n)
        e = np.zeros((parts, n), dtype=np.float64 if npdt in (np.float32, np.float64) else np.uint64)
        e[:, 1:] = np.cumsum(d.reshape(parts, n).astype(e.dtype), axis=1)[:, :-1]
        ok = np.allclose(g.astype(np.float64), (e % 2**32 if npdt in (np.uint32,) else e).astype(np.float64), rtol=1e-5) if npdt != np.int32 else (g.astype(np.int64) == e.astype(np.int64)).all()
        ok_all &= bool(ok)
        if not ok: print("FAIL scan", dt, n, parts)
print("scan ok so far:", ok_all, flush=True)

# reduce
for dt in range(12):
    npdt, comps = O.dtype_info(dt)
    for op in range(4):
        r = G.Reduce(dt, op)
        for n in [1, 5, 31, 100, 1025, 88289, 1 << 20, 5238082]:
            if op == 1:
                d = (rng.integers(0, 3, n * comps) * 0 + 1).astype(npdt); d[rng.integers(0, n * comps, 3)] = 2
            else:
                d = rng.integers(0, 100, n * comps).astype(npdt) if npdt in (np.uint32,) else (rng.random(n * comps) * 200 - 100).astype(npdt)
            b = G.ShaderStorageBuffer(d)
            r(b, n)
            g = b.get_data(npdt)[:comps]
            e = O.reduce_expected(d, dt, op)
            ok = np.allclose(g.astype(np.float64), np.asarray(e).astype(np.float64), rtol=1e-4, atol=1e-1 if op == 0 else 1e-6)
            ok_all &= bool(ok)
            if not ok: print("FAIL reduce", dt, op, n, g, e)
print("reduce ok so far:", ok_all, flush=True)

# timing
for logn in (20, 24, 26, 28):
    n = 1 << logn
    k = rng.integers(0, 2**32, n, dtype=np.uint32); v = np.arange(n, dtype=np.uint32)
    for bits in (4, 8):
        rs = G.RadixSort(digit_bits=bits)
        rs.prepare_internal_buffers(n)
        kb, vb = G.ShaderStorageBuffer(k), G.ShaderStorageBuffer(v)
        k2, v2 = G.ShaderStorageBuffer(size=4 * n), G.ShaderStorageBuffer(size=4 * n)
        times = []
        for it in range(5):
            G.check(G.lib().glu_buffer_copy(kb.handle(), k2.handle(), 4 * n, 0, 0))
            G.check(G.lib().glu_buffer_copy(vb.handle(), v2.handle(), 4 * n, 0, 0))
            ns = G.measure_elapsed_time(lambda: rs(k2, v2, n))
            times.append(ns)
        t = min(times[1:]) * 1e-9
        passes = 32 // bits
        print("N=2^%d bits=%d: %.3f ms  %.1f Mkeys/s  %.0f GB/s(20B/pass x %d)" % (logn, bits, t * 1e3, n / t / 1e6, n * 20 * passes / t / 1e9, passes), flush=True)
        if logn == 28:
            gk = k2.get_data(np.uint32); gv = v2.get_data(np.uint32)
            okk = (np.diff(gk.astype(np.int64)) >= 0).all() and (k[gv] == gk).all()
            print("  2^28 sorted+gather ok:", okk, flush=True)
            ok_all &= bool(okk)
print("ALL OK" if ok_all else "SOME FAILED")
