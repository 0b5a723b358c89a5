"""A plain loop of sorts for kernel traces and counter passes (rocprofv3 ... -- python3 tools/sort_loop.py ...): one sorter object,
prepared, the same pseudo-random input restored before every sort; warm-up sorts first, then `--steps` sorts.
   python tools/sort_loop.py --log2 28 --key-bytes 8 --steps 10 --warmup 3 [--keys-only] [--key-bits B] [--zeros PERCENT] [--distinct K] [--zipf]
Prints the median device time of the timed sorts (library timer) and what glu_radix_sort_read_finish says about the last one."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=float, default=28)
ap.add_argument("--key-bytes", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--key-bits", type=int, default=0, help="keys drawn from [0, 2^B) (0: the whole key)")
ap.add_argument("--zeros", type=float, default=0.0, help="this share of the keys (per cent) is set to zero")
ap.add_argument("--distinct", type=int, default=0, help="the keys are drawn from this many distinct values (0x55555555 * k)")
ap.add_argument("--distinct-scattered", type=int, default=0, help="the keys are drawn from this many distinct pseudo-random values")
ap.add_argument("--zipf", action="store_true", help="Zipf(1.0) over 2^20 values scattered over the key space (tools/measure_distributions_2p28.py)")
ap.add_argument("--keys-only", action="store_true")
ap.add_argument("--digit-bits", type=int, default=8)
a = ap.parse_args()
n = int(round(2 ** a.log2))
rng = np.random.default_rng(0x5EED)
kb = 8 * a.key_bytes
bits = a.key_bits or kb
dt = np.uint64 if a.key_bytes == 8 else np.uint32
keys = rng.integers(0, 2 ** bits, n, dtype=dt)
if a.distinct:
    keys = (rng.integers(0, a.distinct, n, dtype=np.uint32) * np.uint32(0x55555555)).astype(dt)
if a.distinct_scattered:
    pool = rng.integers(0, 2 ** bits, a.distinct_scattered, dtype=dt)
    keys = pool[rng.integers(0, a.distinct_scattered, n)]
if a.zipf:
    ranks = np.floor(np.exp(rng.random(n) * np.log(float(1 << 20)))).astype(np.uint64)
    x = ranks * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    keys = (x >> np.uint64(32)).astype(dt) if a.key_bytes == 4 else x.astype(dt)
if a.zeros > 0:
    keys[rng.random(n) < a.zeros / 100.0] = 0
vals = np.arange(n, dtype=np.uint32)
s = G.RadixSort(digit_bits=a.digit_bits)
s.prepare_internal_buffers(n, key_bytes=a.key_bytes)
k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
k, v = G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)
times = []
for i in range(a.warmup + a.steps):
    G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), keys.nbytes, 0, 0))
    G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), vals.nbytes, 0, 0))
    if a.keys_only:
        t = G.measure_elapsed_time(lambda: s.sort_keys(k, n))
    else:
        t = G.measure_elapsed_time(lambda: s(k, v, n, 0, key_bytes=a.key_bytes))
    if i >= a.warmup:
        times.append(t * 1e-6)
times.sort()
print("n %d  key bytes %d  key bits %d  zeros %.3f %%  median %.3f ms  min %.3f  max %.3f  finish %s" % (
    n, a.key_bytes, bits, a.zeros, times[len(times) // 2], times[0], times[-1], s.read_finish()), s.read_long_runs(), flush=True)
