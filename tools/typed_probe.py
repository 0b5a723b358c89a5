"""Debug: typed sorts through the attempt to end in LDS, every key shape of tools/fuzz.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
os.environ["GLU_HIP_SORT_PAIR_MIN"] = "1"
os.environ["GLU_HIP_SORT_FINISH_MIN"] = "1"
os.environ["GLU_HIP_SORT_FINISH_BACKOFF"] = "0"
import glu_hip as G
rng = np.random.default_rng(7)  # (reseeded per case below)


def draw(kind, n, bits):
    dt = np.uint32 if bits == 32 else np.uint64
    full = rng.integers(0, 2 ** bits, n, dtype=dt)
    if kind == 11:
        m = max(1, n // int(rng.integers(700, 2200)))
        tops = rng.choice(1 << 16, size=min(m, 1 << 16), replace=False).astype(dt)
        pick = tops[rng.integers(0, tops.size, n)]
        return ((full & dt(0xFFFF)) | (pick << dt(bits - 16))).astype(dt)
    if kind == 8:
        b = dt(8 * int(rng.integers(0, bits // 8)))
        hot = rng.random(n) < rng.uniform(0.05, 0.5)
        return np.where(hot, (full & ~(dt(0xFF) << b)) | (dt(int(rng.integers(0, 256))) << b), full).astype(dt)
    if kind == 9:
        b = dt(8 * int(rng.integers(0, bits // 8)))
        rare = rng.random(n) < 1.0 / int(rng.integers(64, 2048))
        return np.where(rare, full | (dt(0x80) << b), full & ~(dt(0x80) << b)).astype(dt)
    if kind == 10:
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
    if kind == 5:
        mask = dt(0)
        for b in range(bits // 8):
            if rng.random() < 0.5:
                mask |= dt(0xFF) << dt(8 * b)
        return (full & mask) | (full[0] & ~mask)
    if kind == 6:
        return full & dt((1 << int(rng.integers(1, bits))) - 1)
    return np.repeat(full[: n // 64 + 1], 64)[:n].copy()


ONLY = os.environ.get("TP_ONLY")
for name in ("float32", "int32", "float64"):
    dt = np.dtype(name)
    for kind in range(12):
        for rep in range(3 if not ONLY else 6):
            if ONLY and ONLY != "%s:%d" % (name, kind):
                continue
            rng = np.random.default_rng(1000 * kind + 10 * rep + dt.itemsize + (1 if dt.kind == "f" else 0))
            n = int(rng.integers(1 << 22, 1 << 23))
            raw = draw(kind, n, dt.itemsize * 8)
            keys = raw.view(dt)
            if dt.kind == "f":
                keys = np.where(np.isnan(keys), dt.type(1.5), keys).astype(dt)
            vals = np.arange(n, dtype=np.uint32)
            s = G.RadixSort()
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, name)
            gk, gv = kb.get_data(dt), vb.get_data(np.uint32)
            u = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
            top = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
            code = (u ^ top) if dt.kind == "i" else np.where(u & top, ~u, u ^ top)
            order = np.argsort(code, kind="stable")
            okk = (gk.view(u.dtype) == u[order]).all()
            okv = (gv == vals[order]).all()
            fin = s.read_finish()
            if not (okk and okv):
                bad = np.flatnonzero(gk.view(u.dtype) != u[order])
                print("FAIL", name, "kind", kind, "n", n, fin, "keys ok", okk, "vals ok", okv, "first bad", bad[:3], "of", bad.size, flush=True)
                exp = u[order]
                got = gk.view(u.dtype)
                for i in bad[:4]:
                    print("   pos %d expected %x got %x  (code of expected %x)" % (i, exp[i], got[i], code[order][i]))
                top16 = (code >> u.dtype.type(dt.itemsize * 8 - 16)).astype(np.int64)
                cnt = np.bincount(top16, minlength=65536)
                runs_of_bad = np.unique(top16[order][bad])
                print("   bad elements lie in %d runs; lengths of the first: %s; bad positions span %d..%d" % (runs_of_bad.size, cnt[runs_of_bad[:8]], bad[0], bad[-1]))
            elif rep == 0:
                print("ok  ", name, "kind", kind, fin, flush=True)
