"""Sort time over sizes with paired passes on / off (from which size is the two-digit table cheaper than a second read of
the keys?): python tools/pairs_ladder.py [pairs|keys|u64|u64keys] [digit bits: 8|4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np, glu_hip as G

mode = sys.argv[1] if len(sys.argv) > 1 else "pairs"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
u64 = mode.startswith("u64")
keys_only = mode.endswith("keys")
os.environ["GLU_HIP_SORT_PAIR_MIN"] = "1"
for log2n in (22, 23, 24, 25, 26, 27, 28):
    for frac in (1.0, 1.41):
        m = int((1 << log2n) * frac)
        if m > (1 << 28):
            continue
        dt = np.uint64 if u64 else np.uint32
        keys = np.random.default_rng(m).integers(0, 2 ** (64 if u64 else 32), m, dtype=dt)
        vals = np.arange(m, dtype=np.uint32)
        k0 = G.ShaderStorageBuffer(keys)
        v0 = G.ShaderStorageBuffer(vals)
        kb = G.ShaderStorageBuffer(size=keys.nbytes)
        vb = G.ShaderStorageBuffer(size=vals.nbytes)
        row = []
        for pairs in ("1", "0"):
            os.environ["GLU_HIP_SORT_PAIRS"] = pairs
            s = G.RadixSort(digit_bits=bits)
            s.prepare_internal_buffers(m, key_bytes=8 if u64 else 4, with_vals=not keys_only)
            best = 1e18
            for r in range(6):
                G.check(G.lib().glu_buffer_copy(k0.handle(), kb.handle(), keys.nbytes, 0, 0))
                G.check(G.lib().glu_buffer_copy(v0.handle(), vb.handle(), vals.nbytes, 0, 0))
                if keys_only:
                    best = min(best, G.measure_elapsed_time(lambda: s.sort_keys_ptr(kb.device_ptr(), m, 0, None, key_bytes=8 if u64 else 4)))
                else:
                    best = min(best, G.measure_elapsed_time(lambda: s(kb, vb, m, 0, key_bytes=8 if u64 else 4)))
            row.append(best * 1e-3)
        print("%-8s %d-bit n %10d (2^%.2f): paired %9.1f us   every pass counts %9.1f us   %+.1f %%" % (
            mode, bits, m, np.log2(m), row[0], row[1], (row[0] / row[1] - 1) * 100), flush=True)
