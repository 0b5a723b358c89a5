"""Sort time by key distribution at 2^28 pairs (uint32 key + uint32 value), at HEAD: does the sort end in LDS, with which run bits, how
many runs go to the segmented passes, how many bytes per pair move.  Every distribution is sorted on a FRESH object first (the first
sort of such keys) and then six more times on the same object (restored input, back to back); the line holds the first sort's time and the median
of the later ones.  The result of the last sort is checked for ascending keys on the host.
   python tools/measure_distributions_2p28.py [log2 pairs = 28]
Record: profiles/r06/distributions_2p28.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
U64 = len(sys.argv) > 2 and sys.argv[2] == "u64"  # python tools/measure_distributions_2p28.py 28 u64: 64-bit keys (profiles/r06/distributions_2p28_u64.txt)
KB = 8 if U64 else 4
KT = np.uint64 if U64 else np.uint32
n = 1 << log2n
rng = np.random.default_rng(2028)
print(G.device_info())
print("2^%d pairs; columns: first sort on a fresh object (ms) | median of 6 later sorts (ms) | ended in LDS | run bits below | tile | runs (pairs) "
      "given to the segmented passes | bytes per pair moved | Gkeys/s (later sorts)" % log2n)


def uniform(bits=32):
    return rng.integers(0, 2 ** bits, n, dtype=np.uint64).astype(np.uint32)


def with_zeros(percent):
    k = uniform()
    k[rng.random(n) < percent / 100.0] = 0
    return k


def zipf_scattered():
    # P(rank k) ~ 1 / k over 2^20 ranks (log-uniform ranks), every rank a pseudo-random 32-bit value: heavy hitters anywhere in the key space
    ranks = np.floor(np.exp(rng.random(n) * np.log(float(1 << 20)))).astype(np.uint64)
    x = ranks * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    return (x >> np.uint64(32)).astype(np.uint32)


def zipf_small_integers():
    return np.floor(np.exp(rng.random(n) * np.log(float(1 << 20)))).astype(np.uint32)  # the ranks themselves: 20-bit keys, rank 1 the most frequent


def uniform64(bits=64):
    return rng.integers(0, 2 ** bits, n, dtype=np.uint64)


def zipf64():
    ranks = np.floor(np.exp(rng.random(n) * np.log(float(1 << 20)))).astype(np.uint64)
    x = ranks * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    return x


def zeros64(percent):
    k = uniform64()
    k[rng.random(n) < percent / 100.0] = 0
    return k


dists64 = [
    ("uniform, full range", lambda: uniform64()),
    ("uniform, 48-bit keys", lambda: uniform64(48)),
    ("uniform, 40-bit keys", lambda: uniform64(40)),
    ("uniform + 0.01 % zeros", lambda: zeros64(0.01)),
    ("uniform + 1 % zeros", lambda: zeros64(1.0)),
    ("Zipf(1.0) over 2^20 values, values scattered", zipf64),
    ("sorted (uniform, ascending)", lambda: np.sort(uniform64())),
    ("three values", lambda: rng.integers(0, 3, n, dtype=np.uint64) * np.uint64(0x5555555555555555)),
    ("1000 distinct values, scattered", lambda: rng.integers(0, 2**64, 1000, dtype=np.uint64)[rng.integers(0, 1000, n)]),
]
dists = [
    ("uniform, full range", lambda: uniform()),
    ("uniform, 31-bit keys", lambda: uniform(31)),
    ("uniform, 30-bit keys", lambda: uniform(30)),
    ("uniform, 28-bit keys", lambda: uniform(28)),
    ("uniform, 26-bit keys", lambda: uniform(26)),
    ("uniform, 24-bit keys", lambda: uniform(24)),
    ("uniform + 0.01 % zeros", lambda: with_zeros(0.01)),
    ("uniform + 1 % zeros", lambda: with_zeros(1.0)),
    ("Zipf(1.0) over 2^20 values, values scattered", zipf_scattered),
    ("Zipf(1.0) over 2^20 values, values = ranks", zipf_small_integers),
    ("sorted (uniform, ascending)", lambda: np.sort(uniform())),
    ("reversed (uniform, descending)", lambda: np.sort(uniform())[::-1].copy()),
    ("three values", lambda: rng.integers(0, 3, n, dtype=np.uint32) * np.uint32(0x55555555)),
    ("100 distinct values, scattered over the key space", lambda: rng.integers(0, 2**32, 100, dtype=np.uint64).astype(np.uint32)[rng.integers(0, 100, n)]),
    ("1000 distinct values, scattered", lambda: rng.integers(0, 2**32, 1000, dtype=np.uint64).astype(np.uint32)[rng.integers(0, 1000, n)]),
    ("2^16 distinct values, scattered", lambda: rng.integers(0, 2**32, 65536, dtype=np.uint64).astype(np.uint32)[rng.integers(0, 65536, n)]),
    ("2^20 distinct values, scattered (256 copies each)", lambda: rng.integers(0, 2**32, 1 << 20, dtype=np.uint64).astype(np.uint32)[rng.integers(0, 1 << 20, n)]),
    ("uniform + 10 % zeros", lambda: with_zeros(10.0)),
    ("all zero (the reference README's benchmark input)", lambda: np.zeros(n, dtype=np.uint32)),
]
if U64:
    dists = dists64
vals = np.arange(n, dtype=np.uint32)
v0 = G.ShaderStorageBuffer(vals)
k = G.ShaderStorageBuffer(size=KB * n)
v = G.ShaderStorageBuffer(size=4 * n)
for name, make in dists:
    keys = make()
    k0 = G.ShaderStorageBuffer(keys)
    s = G.RadixSort()
    s.prepare_internal_buffers(n, key_bytes=KB)
    times = []
    for rep in range(7):  # (back to back: a pause -- a host read-back -- lets the device clock down and costs the next sorts 8 %)
        G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), KB * n, 0, 0))
        G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), 4 * n, 0, 0))
        times.append(G.measure_elapsed_time(lambda: s(k, v, n, 0, key_bytes=KB)) * 1e-6)
    out = k.get_data(KT)
    assert bool((out[1:] >= out[:-1]).all()), name
    del out
    fin, lr = s.read_finish(), s.read_long_runs()
    later = sorted(times[1:])[len(times[1:]) // 2]
    if fin["accepted"]:
        PB = KB + 4  # bytes of a pair
        lp = (2 if KB == 4 else 6) * (2 * PB + KB)  # what the segmented passes move per pair of a long run (less if the run is one key value)
        moved = 2 * 2 * PB + KB + (2 * 256 * 256 * 512 + 2 * 65536 * 4) / n + 2 * PB * (1 - lr["pairs"] / n) + lp * lr["pairs"] / n
        what = "yes  bits [%2d,%2d)  tile %4d  long runs %5d (%9d pairs)" % (fin["top_bit"] - 16, fin["top_bit"], fin["capacity"], lr["runs"], lr["pairs"])
    else:
        NP = KB  # ordinary passes of 8 bits
        skipped, alone, roles = s.read_plan(NP, roles=True)
        reads = sum(1 for p in range(NP) if skipped[p] != 2 and not (roles[p] == 2 and not alone[p])) + (1 if fin["attempted"] else 0)
        moved = 2 * (KB + 4) * sum(1 for p in range(NP) if not skipped[p]) + KB * reads
        what = "no   (%s; %d of %d ordinary passes ran)" % ("refused" if fin["attempted"] else "not attempted", sum(1 for p in range(NP) if not skipped[p]), NP)
    print("%-52s %7.3f | %7.3f | %-72s | %5.1f B | %6.1f" % (name, times[0], later, what, moved, n / later / 1e6), flush=True)
    del k0, keys, s
