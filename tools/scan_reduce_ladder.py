import sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G
print("scan (u32, exclusive +), reduce (u32 sum): device time, best of 12")
n = 1024.0
prev = None
while n <= (1 << 28):
    m = 1 << int(round(np.log2(n)))
    d = np.random.default_rng(0).integers(0, 2**32, m, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    sc = G.BlellochScan(G.DataType_Uint); sc(b, m)
    ts = min(G.measure_elapsed_time(lambda: sc(b, m)) for _ in range(12)) * 1e-9
    rd = G.Reduce(G.DataType_Uint, G.ReduceOperator_Sum); rd(b, m)
    tr = min(G.measure_elapsed_time(lambda: rd(b, m)) for _ in range(12)) * 1e-9
    print("2^%2d  scan %9.1f us %7.0f GB/s   reduce %9.1f us %7.0f GB/s" % (int(np.log2(m)), ts * 1e6, m * 8 / ts / 1e9, tr * 1e6, m * 4 / tr / 1e9), flush=True)
    n *= 2
# partitions
for parts, cnt in ((16, 1 << 18), (256, 1 << 16), (1 << 14, 1 << 10), (1 << 18, 1 << 8)):
    m = parts * cnt
    d = np.random.default_rng(0).integers(0, 2**32, m, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    sc = G.BlellochScan(G.DataType_Uint); sc(b, cnt, parts)
    ts = min(G.measure_elapsed_time(lambda: sc(b, cnt, parts)) for _ in range(12)) * 1e-9
    print("partitions %7d x %7d: scan %9.1f us %7.0f GB/s" % (parts, cnt, ts * 1e6, m * 8 / ts / 1e9), flush=True)
for parts, cnt in ((1 << 20, 1 << 8), (1 << 18, 1 << 10)):  # 2^28 elements without any cross-workgroup carry: what the chip does for the scan's bytes
    m = parts * cnt
    d = np.random.default_rng(0).integers(0, 2**32, m, dtype=np.uint32)
    b = G.ShaderStorageBuffer(d)
    sc = G.BlellochScan(G.DataType_Uint); sc(b, cnt, parts)
    ts = min(G.measure_elapsed_time(lambda: sc(b, cnt, parts)) for _ in range(12)) * 1e-9
    print("partitions %7d x %7d: scan %9.1f us %7.0f GB/s" % (parts, cnt, ts * 1e6, m * 8 / ts / 1e9), flush=True)
