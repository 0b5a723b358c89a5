"""Randomized parity run over the whole C ABI surface (not part of the test suite: open-ended, time-boxed).
Every case draws an entry point, a size (log-uniform up to 2^23, with extra weight next to the geometry switch points), a
key distribution and a digit width, runs it on the GPU and compares with numpy / the oracle.
usage (GPU box): python tools/fuzz.py [seconds] [seed]"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
sys.path.insert(0, ROOT)
import glu_hip as G
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
SWITCH = [1024, 4096, 8192, 12288, 256 * 4096, 768 * 4096, 256 * 12288, 256 * 20480, 256 * 8192, 256 * 2048, 1 << 22,
          256 * 10240 * 3 // 2, 256 * 16384 * 3 // 2, 256 * 8192 * 3 // 2, 256 * 6144 * 3 // 2, 256 * 6144 * 2 + 6144, 256 * 10240 * 2, 256 * 10240 * 2 + 10240,  # + line-kernel switch points
          256 * 4 * 4096 + 1, 3 << 21]  # (round 5) pairs: the line kernel's first size; 64-bit keys: the first size that tries to end in LDS


LARGE = os.environ.get("FUZZ_LARGE") == "1"  # most sizes in [2^22, 2^23], passes paired from 2^22 elements up
if LARGE:
    os.environ["GLU_HIP_SORT_PAIR_MIN"] = "1"  # (default: from 2^28 bytes of keys)
    os.environ["GLU_HIP_SORT_FINISH_MIN"] = "1"     # and every whole-key u32 sort of that size tries to end in LDS (default: from 2^26)
    os.environ["GLU_HIP_SORT_FINISH_BACKOFF"] = "0" # ... whatever the sort before it on the same object was told


def draw_n(limit=1 << 23):
    if LARGE and rng.random() < 0.8:
        return int(rng.integers(1 << 22, limit + 1))
    if rng.random() < 0.3:
        n = int(rng.choice(SWITCH)) + int(rng.integers(-3, 4))
    else:
        n = int(2 ** rng.uniform(0, np.log2(limit)))
    return max(1, min(n, limit))


def draw_keys(n, bits):
    dt = np.uint32 if bits == 32 else np.uint64
    kind = rng.integers(0, 12)
    full = rng.integers(0, 2 ** bits, n, dtype=dt)
    if kind == 11:  # the top 16 bits take few enough values that the runs of equal top bits are about as long as the in-LDS
        # pass's tile (1536 pairs at these sizes): sorts that end in LDS, sorts that are refused, and the border between them
        m = max(1, n // int(rng.integers(700, 2200)))
        tops = rng.choice(1 << 16, size=min(m, 1 << 16), replace=False).astype(dt)
        pick = tops[rng.integers(0, tops.size, n)]
        return ((full & dt(0xFFFF)) | (pick << dt(bits - 16))).astype(dt)
    if kind == 8:  # one byte takes one value in a tenth to a half of the keys (paired passes: units too long to balance)
        b = dt(8 * int(rng.integers(0, bits // 8)))
        hot = rng.random(n) < rng.uniform(0.05, 0.5)
        return np.where(hot, (full & ~(dt(0xFF) << b)) | (dt(int(rng.integers(0, 256))) << b), full).astype(dt)
    if kind == 9:  # half of one byte's values are rare (paired passes: runs of very many tiny units)
        b = dt(8 * int(rng.integers(0, bits // 8)))
        rare = rng.random(n) < 1.0 / int(rng.integers(64, 2048))
        return np.where(rare, full | (dt(0x80) << b), full & ~(dt(0x80) << b)).astype(dt)
    if kind == 10:  # a long prefix of the array shares its low 16 bits (paired passes: 16-bit counter overflow)
        m = int(rng.integers(0, n + 1))
        out = full.copy()
        out[:m] = (out[:m] & ~dt(0xFFFF)) | dt(int(rng.integers(0, 1 << 16)))
        return out
    if kind == 0:
        return full
    if kind == 1:
        return (full % dt(rng.integers(1, 300))).astype(dt)
    if kind == 2:
        return np.sort(full)
    if kind == 3:
        return np.sort(full)[::-1].copy()
    if kind == 4:
        return np.full(n, full[0], dtype=dt)
    if kind == 5:  # some bytes constant
        mask = dt(0)
        for b in range(bits // 8):
            if rng.random() < 0.5:
                mask |= dt(0xFF) << dt(8 * b)
        return (full & mask) | (full[0] & ~mask)
    if kind == 6:
        return full & dt((1 << int(rng.integers(1, bits))) - 1)
    return np.repeat(full[: n // 64 + 1], 64)[:n].copy()  # runs of 64 equal keys


cases = fails = 0
finish_attempts = finish_accepted = 0  # sorts that tried to / did end in LDS (glu_radix_sort_read_finish)
t_end = time.time() + budget
while time.time() < t_end:
    api = str(rng.choice(["pairs", "keys", "u64", "typed", "bits", "scan", "reduce", "steps"]))
    dbits = int(rng.choice([4, 8]))
    desc = None
    try:
        if api in ("pairs", "keys", "steps"):
            n = draw_n()
            keys = draw_keys(n, 32)
            vals = np.arange(n, dtype=np.uint32)
            steps = int(rng.integers(1, 9)) if api == "steps" else 0
            desc = (api, n, dbits, steps)
            assert keys.size == n
            s = G.RadixSort(digit_bits=dbits)
            kb = G.ShaderStorageBuffer(keys)
            field = keys & np.uint32((1 << (4 * steps)) - 1) if steps else keys
            order = np.argsort(field, kind="stable")
            if api == "keys":
                s.sort_keys(kb, n, steps)
                ok = (kb.get_data(np.uint32) == keys[order]).all()
            else:
                vb = G.ShaderStorageBuffer(vals)
                s(kb, vb, n, steps)
                ok = (kb.get_data(np.uint32) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
            fin = s.read_finish()  # (the reads above waited for the sort)
            finish_attempts += fin["attempted"]
            finish_accepted += fin["accepted"]
        elif api == "u64":
            n = draw_n(1 << 23)
            keys = draw_keys(n, 64)
            vals = np.arange(n, dtype=np.uint32)
            desc = (api, n, dbits)
            s = G.RadixSort(digit_bits=dbits)
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s(kb, vb, n, 0, key_bytes=8)
            order = np.argsort(keys, kind="stable")
            ok = (kb.get_data(np.uint64) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
        elif api == "typed":
            n = draw_n(1 << 23)
            name = str(rng.choice(["int32", "float32", "int64", "float64"]))
            dt = np.dtype(name)
            raw = draw_keys(n, dt.itemsize * 8)
            keys = raw.view(dt)
            if dt.kind == "f":
                keys = np.where(np.isnan(keys), dt.type(1.5), keys).astype(dt)
            vals = np.arange(n, dtype=np.uint32)
            desc = (api, name, n, dbits)
            s = G.RadixSort(digit_bits=dbits)
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, name)
            gk, gv = kb.get_data(dt), vb.get_data(np.uint32)
            # total order of the bit patterns: -0 < +0 for floats, so compare through the order-preserving integer code
            u = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
            top = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
            code = (u ^ top) if dt.kind == "i" else np.where(u & top, ~u, u ^ top)
            order = np.argsort(code, kind="stable")
            ok = (gk.view(u.dtype) == u[order]).all() and (gv == vals[order]).all()
        elif api == "bits":
            kbytes = int(rng.choice([4, 8]))
            n = draw_n(1 << 23)
            keys = draw_keys(n, kbytes * 8)
            a, b = sorted(int(x) for x in rng.integers(0, kbytes * 8 + 1, 2))
            vals = np.arange(n, dtype=np.uint32)
            desc = (api, kbytes, n, dbits, a, b)
            dt = keys.dtype
            field = (keys >> dt.type(a)) & dt.type((1 << (b - a)) - 1) if b > a else np.zeros(n, dtype=dt)
            order = np.argsort(field, kind="stable")
            s = G.RadixSort(digit_bits=dbits)
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s.sort_bit_range_ptr(kb.device_ptr(), vb.device_ptr(), n, a, b, None, kbytes)
            ok = (kb.get_data(dt) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
        elif api == "scan":
            count = 1 << int(rng.integers(0, 21))
            parts = int(rng.integers(1, 5))
            d = rng.integers(0, 1000, count * parts, dtype=np.uint32)
            desc = (api, count, parts)
            bfr = G.ShaderStorageBuffer(d)
            G.BlellochScan(G.DataType_Uint)(bfr, count, parts)
            ok = (bfr.get_data(np.uint32) == O.exclusive_scan_u32(d, count, parts)).all()
        else:
            count = draw_n(1 << 22)
            op = int(rng.integers(0, 4))
            d = rng.integers(1, 3 if op == 1 else 2 ** 31, count, dtype=np.uint32)
            desc = (api, count, op)
            bfr = G.ShaderStorageBuffer(d)
            G.Reduce(G.DataType_Uint, op)(bfr, count)
            got = int(bfr.get_data(np.uint32)[0])
            if op == 0:
                exp = int(d.astype(np.uint64).sum() & 0xFFFFFFFF)
            elif op == 1:
                exp = 1
                for x in d[d != 1]:
                    exp = (exp * int(x)) & 0xFFFFFFFF
            else:
                exp = int(d.min() if op == 2 else d.max())
            ok = got == exp
    except Exception as e:  # noqa: BLE001
        ok = False
        desc = (desc, repr(e))
    cases += 1
    if not ok:
        fails += 1
        print("FAIL", desc, flush=True)
print("cases %d, failures %d; sorts that tried to end in LDS %d, that did %d" % (cases, fails, finish_attempts, finish_accepted))
sys.exit(1 if fails else 0)
