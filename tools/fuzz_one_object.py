"""Randomized parity run of sorts that try to end in LDS on ONE sort object (tools/fuzz.py makes a fresh object per case): what a
sort assumes -- the key bits its runs come from, the tile it expects, whether it asks at all -- comes from what the object's
earlier sorts saw, so the sequence matters: key widths, key types, value ranges, skew and sizes change from sort to sort.
usage (GPU box): python tools/fuzz_one_object.py [seconds] [seed]"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
os.environ["GLU_HIP_SORT_PAIR_MIN"] = "1"
os.environ["GLU_HIP_SORT_FINISH_MIN"] = "1"
import glu_hip as G

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)


def draw(n, bits):
    dt = np.uint32 if bits == 32 else np.uint64
    full = rng.integers(0, 2 ** bits, n, dtype=dt)
    kind = int(rng.integers(0, 10))
    if kind == 0:
        return full
    if kind == 7:  # (round 5) several long runs of different lengths among uniform keys: the segmented passes of a sort that ends in LDS
        out = full.copy()
        for r in range(int(rng.integers(2, 40))):
            m = int(rng.integers(1600, 60000))
            pos = rng.choice(n, size=min(m, n), replace=False)
            out[pos] = (out[pos] & dt((1 << (bits - 16)) - 1)) | (dt(int(rng.integers(0, 65536))) << dt(bits - 16))
        return out
    if kind == 8:  # (round 5) a share of one key value among uniform keys
        out = full.copy()
        out[rng.random(n) < float(rng.choice([0.0001, 0.01, 0.2]))] = dt(int(rng.integers(0, 2 ** 31)))
        return out
    if kind == 9:  # (round 5) keys that tie on the bits the in-LDS pass of 64-bit keys ranks: few values in bits [bits - 32, bits - 16)
        few = rng.integers(0, int(rng.integers(1, 300)), n).astype(dt) << dt(bits - 32)
        return (full & ~(dt(0xFFFF) << dt(bits - 32))) | few
    if kind == 1:  # a smaller range, with or without constant bits above it
        k = int(rng.integers(8, bits))
        out = full >> dt(bits - k)
        if rng.random() < 0.3 and k < bits - 1:
            out |= dt(int(rng.integers(1, 2 ** min(bits - k, 16)))) << dt(k)
        return out
    if kind == 2:  # one long run among uniform keys
        out = full.copy()
        m = int(rng.integers(1000, 12000))
        pos = rng.choice(n, size=min(m, n), replace=False)
        out[pos] = (out[pos] & dt((1 << (bits - 16)) - 1)) | (dt(0x1234) << dt(bits - 16))
        return out
    if kind == 3:
        return np.sort(full)
    if kind == 4:  # few distinct keys
        return full % dt(int(rng.integers(1, 1000)))
    if kind == 5:  # half of the runs empty
        return full & ~(dt(1) << dt(bits - 16))
    return np.full(n, full[0], dtype=dt)


cases = fails = attempted = accepted = 0
objects = [G.RadixSort() for _ in range(2)]
t_end = time.time() + budget
while time.time() < t_end:
    s = objects[int(rng.integers(0, len(objects)))]
    n = int(rng.integers(1 << 22, (1 << 23) + (1 << 22)))
    what = str(rng.choice(["u32", "u32", "u32keys", "u64", "int32", "float32", "int64", "float64"]))
    desc = (what, n)
    try:
        vals = np.arange(n, dtype=np.uint32)
        if what in ("u32", "u32keys"):
            keys = draw(n, 32)
            order = np.argsort(keys, kind="stable")
            kb = G.ShaderStorageBuffer(keys)
            if what == "u32keys":
                s.sort_keys(kb, n)
                ok = (kb.get_data(np.uint32) == keys[order]).all()
            else:
                vb = G.ShaderStorageBuffer(vals)
                s(kb, vb, n)
                ok = (kb.get_data(np.uint32) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
        elif what == "u64":
            keys = draw(n, 64)
            order = np.argsort(keys, kind="stable")
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s(kb, vb, n, 0, key_bytes=8)
            ok = (kb.get_data(np.uint64) == keys[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
        else:
            dt = np.dtype(what)
            raw = draw(n, dt.itemsize * 8)
            keys = raw.view(dt)
            if dt.kind == "f":
                keys = np.where(np.isnan(keys), dt.type(1.5), keys).astype(dt)
            kb, vb = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
            s.sort_typed_ptr(kb.device_ptr(), vb.device_ptr(), n, what)
            u = keys.view(np.uint32 if dt.itemsize == 4 else np.uint64)
            top = u.dtype.type(1) << u.dtype.type(dt.itemsize * 8 - 1)
            code = (u ^ top) if dt.kind == "i" else np.where(u & top, ~u, u ^ top)
            order = np.argsort(code, kind="stable")
            ok = (kb.get_data(dt).view(u.dtype) == u[order]).all() and (vb.get_data(np.uint32) == vals[order]).all()
        fin = s.read_finish()
        attempted += fin["attempted"]
        accepted += fin["accepted"]
        cases += 1
        if not ok:
            fails += 1
            print("FAIL", desc, fin, flush=True)
    except Exception as e:  # noqa
        fails += 1
        print("FAIL", desc, repr(e), flush=True)
print("cases %d, failures %d; sorts that tried to end in LDS %d, that did %d" % (cases, fails, attempted, accepted))
