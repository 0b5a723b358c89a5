"""Does it matter WHERE the caller's key / value arrays lie?  One prepared sorter (its scratch placed by measurement), P caller pairs
allocated one after the other, the same pseudo-random input copied into each before every sort: median device time per pair.
   python tools/caller_pairs_probe.py [--log2 28] [--key-bytes 4] [--pairs 12]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=float, default=28)
ap.add_argument("--key-bytes", type=int, default=4)
ap.add_argument("--pairs", type=int, default=12)
a = ap.parse_args()
n = int(round(2 ** a.log2))
rng = np.random.default_rng(0x5EED)
dt = np.uint64 if a.key_bytes == 8 else np.uint32
keys = rng.integers(0, 2 ** (8 * a.key_bytes), n, dtype=dt)
vals = np.arange(n, dtype=np.uint32)
s = G.RadixSort()
s.prepare_internal_buffers(n, key_bytes=a.key_bytes)
print("placement:", s.scratch_placement() if hasattr(s, "scratch_placement") else "?", flush=True)
k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
pairs = [(G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)) for _ in range(a.pairs)]
for rnd in range(2):
    out = []
    for k, v in pairs:
        t = []
        for rep in range(4):
            G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), keys.nbytes, 0, 0))
            G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), vals.nbytes, 0, 0))
            t.append(G.measure_elapsed_time(lambda: s(k, v, n, 0, key_bytes=a.key_bytes)) * 1e-6)
        out.append(sorted(t[1:])[1])
    print("round %d: per caller pair (ms): %s   spread %.3f .. %.3f" % (rnd, " ".join("%.3f" % x for x in out), min(out), max(out)), flush=True)
