import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gl-radix-sort_amd"))
os.environ["GLU_VERBOSE"] = "1"
import glu_hip as G
G.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")
def free():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value
f0 = free()
d = G.Dist(G.dist_unique_id(), 1, 0)
f1 = free(); print("after create: held MiB", (f0 - f1) >> 20, flush=True)
n = 1 << 27
d.prepare(n, 0)
f2 = free(); print("after prepare(n, 0): held MiB", (f0 - f2) >> 20, flush=True)
d.prepare(n, n + 4096)
f3 = free(); print("after prepare(n, n+4096): held MiB", (f0 - f3) >> 20, flush=True)
for i in range(1, 5):
    d.prepare(n + 8192 * i, n + 4096 + 8192 * i)
    print("after re-prepare %d (8192 pairs more each): held MiB" % i, (f0 - free()) >> 20, flush=True)
s = G.RadixSort()
for i in range(4):
    s.prepare_internal_buffers(n + 8192 * i)
    print("plain sorter, prepare %d: held MiB" % i, (f0 - free()) >> 20, flush=True)
