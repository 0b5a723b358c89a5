"""Sort time against the distance between the caller's key and value arrays when both are cut from ONE allocation:
keys at the start, values at 1 GiB + d.   python tools/placement_offsets.py"""
import sys
sys.path.insert(0, "gl-radix-sort_amd")
import numpy as np, glu_hip as G

n = 1 << 28
GiB, MiB = 1 << 30, 1 << 20
keys = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint32)
vals = np.arange(n, dtype=np.uint32)
k0, v0 = G.ShaderStorageBuffer(keys), G.ShaderStorageBuffer(vals)
pool = G.ShaderStorageBuffer(size=3 * GiB + 512 * MiB)
s = G.RadixSort(); s.prepare_internal_buffers(n)
base = pool.device_ptr()
print("pool at %x" % base)
def d2d(dst, src_buffer, nbytes):  # through the library's own runtime instance
    w = G.ShaderStorageBuffer.wrap(dst, nbytes)
    G.check(G.lib().glu_buffer_copy(src_buffer.handle(), w.handle(), nbytes, 0, 0))
for rep in range(2):
    for d in (0, 64, 256, 512, 768, 1024):
        kp, vp = base, base + GiB + d * MiB
        t = 1e9
        for _ in range(2):
            d2d(kp, k0, 4 * n); d2d(vp, v0, 4 * n)
            t = min(t, G.measure_elapsed_time(lambda: s.run_ptr(kp, vp, n, 0, None)) * 1e-6)
        print("rep %d  vals - keys = 1 GiB + %4d MiB: %.3f ms" % (rep, d, t), flush=True)
print("separately allocated caller arrays, same sorter:")
for rep in range(6):
    kb, vb = G.ShaderStorageBuffer(size=4 * n), G.ShaderStorageBuffer(size=4 * n)
    t = 1e9
    for _ in range(2):
        G.check(G.lib().glu_buffer_copy(k0.handle(), kb.handle(), 4 * n, 0, 0)); G.check(G.lib().glu_buffer_copy(v0.handle(), vb.handle(), 4 * n, 0, 0))
        t = min(t, G.measure_elapsed_time(lambda: s(kb, vb, n)) * 1e-6)
    print("  keys %x vals %x (distance %.1f MiB): %.3f ms" % (kb.device_ptr(), vb.device_ptr(), (vb.device_ptr() - kb.device_ptr()) / MiB, t), flush=True)
    keepalive = globals().setdefault("keepalive", []); keepalive.append((kb, vb))
