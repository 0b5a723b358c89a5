"""Randomized parity run for sorts whose keys crowd (round 6): sizes from 2^24 to 2^26.5 pairs -- where a block of the leader's count
kernel is long enough for a heavy hitter to wrap a 16-bit counter of the two-digit table (radix_pair_passes.hpp: wide rows) --,
a few heavy hitters with random shares over uniform / small-range / few-distinct backgrounds, every key kind.  Not part of the test
suite: open-ended, time-boxed.  The result is checked by properties that are complete and cheap at these sizes: the keys equal
np.sort of the input; every output value indexes an input key equal to the output key beside it; values ascend inside a run of
equal keys (stable, and hence a permutation).
usage (GPU box): python tools/fuzz_heavy.py [seconds] [seed]"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import glu_hip as G

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)


def draw_keys(n, bits):
    dt = np.uint32 if bits == 32 else np.uint64
    background = int(rng.integers(0, 5))
    if background == 0:
        keys = rng.integers(0, 2 ** bits, n, dtype=dt)
    elif background == 1:  # a smaller range (the device chooses the runs' bits)
        keys = rng.integers(0, 2 ** int(rng.integers(17, bits)), n, dtype=dt)
    elif background == 2:  # few distinct values, scattered over the key space
        pool = rng.integers(0, 2 ** bits, int(rng.integers(2, 3000)), dtype=dt)
        keys = pool[rng.integers(0, pool.size, n)]
    elif background == 3:  # few distinct top-16 values, every low bit random: long runs that are not one value
        tops = rng.integers(0, 1 << 16, int(rng.integers(1, 400)), dtype=dt)
        keys = (rng.integers(0, 2 ** (bits - 16), n, dtype=dt)) | (tops[rng.integers(0, tops.size, n)] << dt(bits - 16))
    else:  # sorted
        keys = np.sort(rng.integers(0, 2 ** bits, n, dtype=dt))
    # heavy hitters: 0 .. 20 values (more than the 14 wide rows a block can put right: some sorts must be refused and still be right)
    hitters = int(rng.choice([0, 1, 1, 2, 3, 5, 15, 20]))
    shares = rng.dirichlet(np.ones(hitters + 1)) * rng.uniform(0.2, 1.0) if hitters else []
    where = rng.random(n)
    lo = 0.0
    for h in range(hitters):
        value = dt(int(rng.integers(0, 2 ** bits, dtype=np.uint64) if bits == 64 else rng.integers(0, 2 ** bits)))
        if rng.random() < 0.3:
            value = dt(int(value) & ~0xFFFF)  # (shares its top bits with other keys' runs less often than its low ones)
        hit = (where >= lo) & (where < lo + shares[h])
        if rng.random() < 0.25:  # a contiguous stretch instead of a scattering: whole blocks of one value
            a = int(rng.integers(0, n))
            hit = np.zeros(n, dtype=bool)
            hit[a:a + int(shares[h] * n)] = True
        keys[hit] = value
        lo += shares[h]
    return keys


cases = fails = accepted = refused = 0
t_end = time.time() + budget
while time.time() < t_end:
    kind = str(rng.choice(["pairs", "pairs", "keys", "u64", "int32", "float32", "int64"]))
    n = int(2 ** rng.uniform(24.0, 26.5)) + int(rng.integers(0, 4096))
    if kind in ("u64", "int64"):
        n = min(n, 1 << 26)
    desc = (kind, n)
    try:
        bits = 64 if kind in ("u64", "int64") else 32
        raw = draw_keys(n, bits)
        vals = np.arange(n, dtype=np.uint32)
        s = G.RadixSort()
        if kind in ("pairs", "keys", "u64"):
            kb = G.ShaderStorageBuffer(raw)
            if kind == "keys":
                s.sort_keys(kb, n)
                gv = None
            else:
                vb = G.ShaderStorageBuffer(vals)
                s(kb, vb, n, 0, key_bytes=bits // 8)
                gv = vb.get_data(np.uint32)
            gk = kb.get_data(raw.dtype)
            code_in, code_out = raw, gk
        else:
            dt = np.dtype(kind)
            keys = raw.view(dt)
            if dt.kind == "f":
                keys = np.where(np.isnan(keys), dt.type(1.5), keys).astype(dt)
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, kind)
            gk, gv = kb.get_data(dt), vb.get_data(np.uint32)

            def code(a):  # the order-preserving integer code of the bit patterns (-0 < +0)
                u = a.view(np.uint32 if dt.itemsize == 4 else np.uint64)
                top = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
                return (u ^ top) if dt.kind == "i" else np.where(u & top, ~u, u ^ top)
            code_in, code_out = code(keys), code(gk)
        ok = bool((code_out == np.sort(code_in)).all())
        if ok and gv is not None:
            ok = bool((code_in[gv] == code_out).all())
            same = code_out[1:] == code_out[:-1]
            ok = ok and bool((gv[1:][same] > gv[:-1][same]).all())
        fin = s.read_finish()
        accepted += fin["accepted"]
        refused += fin["attempted"] - fin["accepted"]
    except Exception as e:  # noqa: BLE001
        ok = False
        print("EXCEPTION", desc, repr(e), flush=True)
    cases += 1
    if not ok:
        fails += 1
        print("FAIL", desc, "seed", seed, "case", cases, flush=True)
    del s
print("cases %d  failures %d  (ended in LDS %d, refused %d)  seed %d" % (cases, fails, accepted, refused, seed), flush=True)
sys.exit(1 if fails else 0)
