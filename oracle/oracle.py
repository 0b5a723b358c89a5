"""ctypes/numpy front end of the CPU oracle (oracle/glu_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of glu_oracle.c.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libglu_oracle.so")
_lib = None


def build(force=False):
    """Compile libglu_oracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "glu_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libglu_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u64, u32, i32, p = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p
        L.glu_oracle_div_ceil.restype = u64
        L.glu_oracle_div_ceil.argtypes = [u64, u64]
        L.glu_oracle_is_power_of_2.restype = i32
        L.glu_oracle_is_power_of_2.argtypes = [u64]
        L.glu_oracle_next_power_of_2.restype = u64
        L.glu_oracle_next_power_of_2.argtypes = [u64]
        L.glu_oracle_minstd_sample.restype = None
        L.glu_oracle_minstd_sample.argtypes = [u64, u64, u32, u32, p]
        L.glu_oracle_blelloch_scan_u32.restype = i32
        L.glu_oracle_blelloch_scan_u32.argtypes = [p, u64, u64]
        L.glu_oracle_exclusive_scan_u32.restype = None
        L.glu_oracle_exclusive_scan_u32.argtypes = [p, p, u64, u64]
        L.glu_oracle_radix_block_count_buffer_size.restype = u64
        L.glu_oracle_radix_block_count_buffer_size.argtypes = [u64]
        L.glu_oracle_radix_scratch_buffer_size.restype = u64
        L.glu_oracle_radix_scratch_buffer_size.argtypes = [u64]
        L.glu_oracle_radix_sort_reference.restype = i32
        L.glu_oracle_radix_sort_reference.argtypes = [p, p, p, p, u64, u64, p, ctypes.POINTER(i32)]
        L.glu_oracle_stable_sort_pairs_u32.restype = i32
        L.glu_oracle_stable_sort_pairs_u32.argtypes = [p, p, u64, u32]
        L.glu_oracle_stable_sort_pairs_u64.restype = i32
        L.glu_oracle_stable_sort_pairs_u64.argtypes = [p, p, u64, u32]
        L.glu_oracle_reduce_reference_u32.restype = i32
        L.glu_oracle_reduce_reference_u32.argtypes = [p, u64, i32, u32]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def minstd_sample(seed, n, lo, hi):
    """glu::Random(seed).sample_int_vector<GLuint>(n, lo, hi)  (test/util/Random.hpp:32-39)."""
    out = np.empty(n, dtype=np.uint32)
    lib().glu_oracle_minstd_sample(seed, n, lo, hi, _ptr(out))
    return out


def blelloch_scan_u32(data, count, num_partitions=1):
    """Literal BlellochScan::operator() on a copy of `data`; returns the scanned array."""
    d = np.ascontiguousarray(data, dtype=np.uint32).copy()
    rc = lib().glu_oracle_blelloch_scan_u32(_ptr(d), count, num_partitions)
    if rc != 0:
        raise ValueError("BlellochScan argument check failed (reference would exit(1))")
    return d


def exclusive_scan_u32(data, count, num_partitions=1):
    d = np.ascontiguousarray(data, dtype=np.uint32)
    out = np.empty_like(d)
    lib().glu_oracle_exclusive_scan_u32(_ptr(d), _ptr(out), count, num_partitions)
    return out


def radix_sort_reference(keys, vals, num_steps=0, trace=False):
    """Literal RadixSort::operator().

    Returns a dict: user_keys/user_vals (the caller's buffers afterwards), scratch_keys/scratch_vals,
    passes, result_in_scratch, result_keys/result_vals (wherever the reference left the last pass
    output) and, with trace=True, `tables` = the scanned block-count table of each pass.
    """
    k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
    v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
    n = k.size
    ks = np.zeros(max(n, 1), dtype=np.uint32)
    vs = np.zeros(max(n, 1), dtype=np.uint32)
    tables = None
    tptr = None
    if trace and n > 1:
        nb = int(lib().glu_oracle_div_ceil(n, 1024))
        nbp2 = int(lib().glu_oracle_next_power_of_2(nb))
        tables = np.zeros((8, 16 * nbp2), dtype=np.uint32)
        tptr = _ptr(tables)
    in_scratch = ctypes.c_int(0)
    passes = lib().glu_oracle_radix_sort_reference(_ptr(k), _ptr(v), _ptr(ks), _ptr(vs), n, num_steps, tptr,
                                                   ctypes.byref(in_scratch))
    res = {
        "user_keys": k, "user_vals": v, "scratch_keys": ks[:n], "scratch_vals": vs[:n],
        "passes": passes, "result_in_scratch": bool(in_scratch.value),
    }
    res["result_keys"] = res["scratch_keys"] if in_scratch.value else k
    res["result_vals"] = res["scratch_vals"] if in_scratch.value else v
    if tables is not None:
        res["tables"] = tables[:max(passes, 0)]
    return res


def stable_sort_pairs(keys, vals, key_bits=None):
    """Stable sort of (key, val) by the low key_bits bits of key; uint32 or uint64 keys."""
    keys = np.ascontiguousarray(keys)
    v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
    if keys.dtype == np.uint64:
        k = keys.copy()
        rc = lib().glu_oracle_stable_sort_pairs_u64(_ptr(k), _ptr(v), k.size, 64 if key_bits is None else key_bits)
    else:
        k = keys.astype(np.uint32, copy=True)
        rc = lib().glu_oracle_stable_sort_pairs_u32(_ptr(k), _ptr(v), k.size, 32 if key_bits is None else key_bits)
    if rc != 0:
        raise MemoryError("oracle stable sort failed")
    return k, v


def reduce_reference_u32(data, count, op=0, subgroup_size=32):
    """Literal Reduce::operator() for uint32; returns data[0] afterwards."""
    d = np.ascontiguousarray(data, dtype=np.uint32).copy()
    rc = lib().glu_oracle_reduce_reference_u32(_ptr(d), count, op, subgroup_size)
    if rc != 0:
        raise ValueError("Reduce argument check failed (reference would exit(1))")
    return int(d[0])


# numpy statements of what the reference's tests compare against, for every DataType x ReduceOperator
_NP_DTYPES = {
    0: (np.float32, 1), 1: (np.float64, 1), 2: (np.int32, 1), 3: (np.uint32, 1),
    4: (np.float32, 2), 5: (np.float32, 4), 6: (np.float64, 2), 7: (np.float64, 4),
    8: (np.uint32, 2), 9: (np.uint32, 4), 10: (np.int32, 2), 11: (np.int32, 4),
}


def dtype_info(data_type):
    """(numpy scalar dtype, components) of a glu::DataType value (glu/data_types.hpp:8-22)."""
    return _NP_DTYPES[data_type]


def reduce_expected(data, data_type, op):
    """Component-wise reduction the way the reference's tests state it (std::accumulate / min / max);
    float types in float64 so a tolerance test has an accurate centre."""
    dt, comps = _NP_DTYPES[data_type]
    a = np.asarray(data, dtype=dt).reshape(-1, comps)
    if np.issubdtype(dt, np.floating):
        a = a.astype(np.float64)
        r = {0: a.sum(0), 1: a.prod(0), 2: a.min(0), 3: a.max(0)}[op]
        return r
    # integers wrap modulo 2^32 like GLSL
    if op == 0:
        r = a.astype(np.uint64).sum(0) & 0xFFFFFFFF
        return r.astype(np.uint32).view(dt) if dt == np.int32 else r.astype(np.uint32)
    if op == 1:
        r = np.ones(comps, dtype=np.uint64)
        for row in a.astype(np.uint32).astype(np.uint64):
            r = (r * row) & 0xFFFFFFFF
        return r.astype(np.uint32).view(dt) if dt == np.int32 else r.astype(np.uint32)
    return a.min(0) if op == 2 else a.max(0)
