"""Times the BASELINE.json configurations that fit one GPU (device time via the library timer on restored inputs).
Configurations of 2^26 elements and more are timed on ten PLACEMENTS of their arrays (five caller pairs x two sorter
objects) and every line leads with the MEDIAN of the ten, minimum and maximum in brackets: where the arrays lie in HBM moves
these numbers by 5-8 %, and the median is what a caller gets.  Smaller ones: best of 5.
python tools/measure_configs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import numpy as np
import glu_hip as G

print(G.device_info())
rng = np.random.default_rng(0x5EED)


def time_sort(keys, vals, bits, key_bytes=4, reps=5):
    """Device time over `placements`: the caller's arrays are allocated 5 times and the sorter (its scratch) twice for the
    2^28 configurations, because where the arrays lie in HBM decides between discrete speeds of the scatter kernel (C5:
    1.31 / 1.39 / 1.50 ms per pass).  Returns the MEDIAN over the placements (the best of `reps` for
    the small configurations, which have one placement); time_sort.spread = (min, median, max, placements)."""
    n = keys.size
    many = n >= 1 << 26
    sorters = []
    for _ in range(2 if many else 1):
        s = G.RadixSort(digit_bits=bits)
        s.prepare_internal_buffers(n, key_bytes=key_bytes)
        sorters.append(s)
    k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
    targets = [(G.ShaderStorageBuffer(size=keys.nbytes), G.ShaderStorageBuffer(size=vals.nbytes)) for _ in range(5 if many else 1)]
    times = []
    for s in sorters:
        for k, v in targets:
            t_here = 1e18
            for _ in range(1 if many else reps):
                G.check(G.lib().glu_buffer_copy(k0.handle(), k.handle(), keys.nbytes, 0, 0))
                G.check(G.lib().glu_buffer_copy(v0.handle(), v.handle(), vals.nbytes, 0, 0))
                t_here = min(t_here, G.measure_elapsed_time(lambda: s(k, v, n, 0, key_bytes=key_bytes)))
            times.append(t_here)
    times.sort()
    best = times[len(times) // 2] if many else times[0]
    time_sort.spread = (times[0] * 1e-6, times[len(times) // 2] * 1e-6, times[-1] * 1e-6, len(times))
    s = sorters[0]
    # bytes really moved per pair: a pass whose count table came from its leader's two-digit histogram did not read the keys
    # a second time (the pair moved 64 MiB of tables instead)
    moved = None
    time_sort.key_reads = None
    if n >= 1 << 22:
        passes = (8 * key_bytes) // bits
        skipped, alone, roles = s.read_plan(passes, roles=True)
        from_table = sum(1 for p in range(passes) if roles[p] == 2 and not alone[p] and skipped[p] != 2)
        # count kernels that read the keys: not those of passes known to be identities before counting (skip value 2)
        time_sort.key_reads = sum(1 for p in range(passes) if skipped[p] != 2 and not (roles[p] == 2 and not alone[p]))
        leaders = sum(1 for p in range(passes) if roles[p] == 1)
        tables = 2 * 256 * 256 * 512 if bits == 8 else 2 * (256 * 16 * 1024 + 16 * 256 * 16 * 4)  # written and read once
        moved = passes * 2 * (key_bytes + 4) + (passes - from_table) * key_bytes + leaders * tables / n
        fin = s.read_finish()
        if fin["accepted"]:
            # the sort ended in LDS (glu_radix_sort_read_finish): two counting passes on the top 16 bits with one read of the
            # keys and one two-digit table, then one pass that reads and writes every pair once
            time_sort.key_reads = 1
            moved = 2 * 2 * (key_bytes + 4) + key_bytes + (2 * 256 * 256 * 512 + 2 * 65536 * 4) / n + 2 * (key_bytes + 4)
    return best * 1e-9, moved


rows = []
for name, log2n, kind, key_bytes in (("C2 2^20 u32+u32 uniform", 20, "uniform", 4), ("C3 2^28 u32+u32 uniform", 28, "uniform", 4),
                                     ("   2^28 u32+u32 all-zero keys (README input)", 28, "zero", 4),
                                     ("C5 2^28 u64+u32 uniform", 28, "uniform", 8)):
    n = 1 << log2n
    if key_bytes == 4:
        keys = rng.integers(0, 2**32, n, dtype=np.uint32) if kind == "uniform" else np.zeros(n, dtype=np.uint32)
    else:
        keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    for bits in (8, 4):
        t, moved = time_sort(keys, vals, bits, key_bytes)
        passes = (8 * key_bytes) // bits
        bpp = passes * (3 * key_bytes + 8)
        if kind == "zero":
            # every pass has a constant digit and is skipped on the device: only the count kernels read the keys
            rd = time_sort.key_reads * key_bytes
            print("%-46s digits %d-bit: %8.3f ms  %9.1f Mkeys/s  (all passes skipped, %d of them known to be identities before counting: %d B/pair read, %.0f GB/s)" % (
                name, bits, t * 1e3, n / t / 1e6, passes - time_sort.key_reads, rd, n * rd / t / 1e9), flush=True)
            continue
        own = "" if moved is None or abs(moved - bpp) < 0.01 else "; moved %.1f B/pair: %.1f %%" % (moved, n * moved / t / 8e12 * 100)
        sp = time_sort.spread
        spread = "" if sp[3] == 1 else "  [median of %d placements; min %.3f max %.3f ms]" % (sp[3], sp[0], sp[2])
        print("%-46s digits %d-bit: %8.3f ms  %9.1f Mkeys/s  %6.0f GB/s at %d B/pair (%.1f %% of 8 TB/s%s)%s" % (
            name, bits, t * 1e3, n / t / 1e6, n * bpp / t / 1e9, bpp, n * bpp / t / 8e12 * 100, own, spread), flush=True)

n = 1 << 28
d = rng.integers(0, 2**32, n, dtype=np.uint32)
b = G.ShaderStorageBuffer(d)
sc = G.BlellochScan(G.DataType_Uint)
sc(b, n)
ts = sorted(G.measure_elapsed_time(lambda: sc(b, n)) for _ in range(9))
t = ts[len(ts) // 2] * 1e-9
print("BlellochScan 2^28 u32: %.3f ms  %.0f GB/s at 8 B/elem (%.1f %%)  [median of 9 runs; min %.3f max %.3f ms]" % (
    t * 1e3, n * 8 / t / 1e9, n * 8 / t / 8e12 * 100, ts[0] * 1e-6, ts[-1] * 1e-6))
rd = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum)
rd(b, n)
ts = sorted(G.measure_elapsed_time(lambda: rd(b, n)) for _ in range(9))
t = ts[len(ts) // 2] * 1e-9
print("Reduce 2^28 u32 sum:   %.3f ms  %.0f GB/s at 4 B/elem (%.1f %%)  [median of 9 runs; min %.3f max %.3f ms]" % (
    t * 1e3, n * 4 / t / 1e9, n * 4 / t / 8e12 * 100, ts[0] * 1e-6, ts[-1] * 1e-6))
