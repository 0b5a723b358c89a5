"""ctypes binding of libglu_hip.so (include/glu_hip.h) -- used by the tests, bench.py and the
multi-GPU driver.  The C++17 headers in ../glu/ are the drop-in surface for the reference's users; this
module is the thinnest possible Python view of the same C ABI.

There is no CPU fallback: importing works anywhere (so that symbol/ABI tests run without a GPU), but every
compute call returns GLU_ERROR_NO_DEVICE without an MI355X and raises GluError.
"""
import ctypes
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("GLU_HIP_LIB_PATH") or os.path.join(_PKG_DIR, "lib", "libglu_hip.so")  # override: tuning builds

# glu/data_types.hpp:8-22 and glu/Reduce.hpp:42-48 (same numeric values)
DataType_Float, DataType_Double, DataType_Int, DataType_Uint, DataType_Vec2, DataType_Vec4, DataType_DVec2, \
    DataType_DVec4, DataType_UVec2, DataType_UVec4, DataType_IVec2, DataType_IVec4 = range(12)
ReduceOperator_Sum, ReduceOperator_Mul, ReduceOperator_Min, ReduceOperator_Max = range(4)

GLU_OK = 0
GLU_ERROR_INVALID_ARGUMENT = 1
GLU_ERROR_INVALID_STATE = 2
GLU_ERROR_OUT_OF_MEMORY = 3
GLU_ERROR_DEVICE = 4
GLU_ERROR_NO_DEVICE = 5

# every symbol include/glu_hip.h declares: (name, restype, argtypes)
_u32, _u64, _sz, _int, _vp = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p
_P = ctypes.POINTER
SYMBOLS = [
    ("glu_last_error", ctypes.c_char_p, []),
    ("glu_version", ctypes.c_char_p, []),
    ("glu_device_count", _int, [_P(_int)]),
    ("glu_set_device", _int, [_int]),
    ("glu_device_info", _int, [ctypes.c_char_p, _sz]),
    ("glu_device_synchronize", _int, []),
    ("glu_queue", _int, [_P(_vp)]),
    ("glu_buffer_create", _int, [_sz, _P(_u32)]),
    ("glu_buffer_create_with_data", _int, [_vp, _sz, _P(_u32)]),
    ("glu_buffer_wrap", _int, [_vp, _sz, _P(_u32)]),
    ("glu_buffer_destroy", _int, [_u32]),
    ("glu_buffer_size", _int, [_u32, _P(_sz)]),
    ("glu_buffer_device_ptr", _int, [_u32, _P(_vp)]),
    ("glu_buffer_write", _int, [_u32, _vp, _sz, _sz]),
    ("glu_buffer_read", _int, [_u32, _vp, _sz, _sz]),
    ("glu_buffer_fill_u32", _int, [_u32, _u32]),
    ("glu_buffer_copy", _int, [_u32, _u32, _sz, _sz, _sz]),
    ("glu_radix_sort_create", _int, [_P(_vp)]),
    ("glu_radix_sort_destroy", _int, [_vp]),
    ("glu_radix_sort_prepare", _int, [_vp, _sz]),
    ("glu_radix_sort_prepare_u64", _int, [_vp, _sz]),
    ("glu_radix_sort_prepare_ex", _int, [_vp, _sz, _sz, _int]),
    ("glu_radix_sort_run", _int, [_vp, _u32, _u32, _sz, _sz]),
    ("glu_radix_sort_run_ptr", _int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    ("glu_radix_sort_run_u64", _int, [_vp, _u32, _u32, _sz, _sz]),
    ("glu_radix_sort_run_u64_ptr", _int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    ("glu_radix_sort_run_keys", _int, [_vp, _u32, _sz, _sz]),
    ("glu_radix_sort_run_keys_ptr", _int, [_vp, _vp, _sz, _sz, _vp]),
    ("glu_radix_sort_run_keys_u64_ptr", _int, [_vp, _vp, _sz, _sz, _vp]),
    ("glu_radix_sort_run_typed_ptr", _int, [_vp, _vp, _vp, _sz, _int, _vp]),
    ("glu_radix_sort_run_bit_range_ptr", _int, [_vp, _vp, _vp, _sz, _u32, _u32, _u32, _vp]),
    ("glu_radix_sort_partition_ptr", _int, [_vp, _vp, _vp, _vp, _vp, _sz, _u32, _u32, _vp, _vp]),
    ("glu_radix_sort_run_segments_ptr", _int, [_vp, _vp, _vp, _vp, _vp, _sz, _P(_u64), _P(_u64), _P(_u32), _sz, _u32, _u32, _vp]),
    ("glu_radix_sort_plan_segments", _int, [_P(_u64), _P(_u64), _P(_u32), _sz, _u32, _u32, _P(_u32), _sz, _P(_u32), _P(_u32), _P(_u64), _P(_sz)]),
    ("glu_radix_sort_set_digit_bits", _int, [_vp, _u32]),
    ("glu_radix_sort_set_option", _int, [_vp, ctypes.c_char_p, ctypes.c_longlong]),
    ("glu_radix_sort_get_digit_bits", _int, [_vp, _P(_u32)]),
    ("glu_radix_sort_scratch_size", _int, [_vp, _P(_sz)]),
    ("glu_radix_sort_scratch_placement", _int, [_vp, _P(_u32), _P(ctypes.c_double), _P(ctypes.c_double)]),
    ("glu_radix_sort_set_profiling", _int, [_vp, _int]),
    ("glu_radix_sort_read_profile", _int, [_vp, _P(ctypes.c_double), _P(ctypes.c_double), _P(ctypes.c_double), _P(_u64)]),
    ("glu_radix_sort_read_plan", _int, [_vp, _P(_u32), _P(_u32), _P(_u32), _sz]),
    ("glu_radix_sort_read_profile_finish", _int, [_vp, _P(ctypes.c_double), _P(ctypes.c_double), _P(ctypes.c_double), _P(_u64),
                                                  _P(ctypes.c_double), _P(_u64)]),
    ("glu_radix_sort_plan_finish", _int, [_sz, _u32, _P(_u32), _P(_u32)]),
    ("glu_radix_sort_read_finish", _int, [_vp, _P(_u32), _P(_u32), _P(_u32), _P(_u32), _P(_u32)]),
    ("glu_radix_sort_read_long_runs", _int, [_vp, _P(_u32), _P(_u32), _P(_u32)]),
    ("glu_radix_sort_read_seg_finish", _int, [_vp, _P(_u32), _P(_u32), _P(_u32), _P(_u32), _P(_u32), _P(_u32), _P(_u32)]),
    ("glu_scan_create", _int, [_int, _P(_vp)]),
    ("glu_scan_destroy", _int, [_vp]),
    ("glu_scan_prepare", _int, [_vp, _sz, _sz]),
    ("glu_scan_run", _int, [_vp, _u32, _sz, _sz]),
    ("glu_scan_run_ptr", _int, [_vp, _vp, _sz, _sz, _vp]),
    ("glu_reduce_create", _int, [_int, _int, _P(_vp)]),
    ("glu_reduce_destroy", _int, [_vp]),
    ("glu_reduce_run", _int, [_vp, _u32, _sz]),
    ("glu_reduce_run_ptr", _int, [_vp, _vp, _sz, _vp]),
    ("glu_dist_available", _int, []),
    ("glu_dist_unique_id", _int, [_vp, _sz]),
    ("glu_dist_create", _int, [_vp, _sz, _int, _int, _P(_vp)]),
    ("glu_dist_destroy", _int, [_vp]),
    ("glu_dist_world", _int, [_vp, _P(_int), _P(_int)]),
    ("glu_dist_partition_shift", _int, [_vp, _P(_u32)]),
    ("glu_dist_local_sorter", _int, [_vp, _P(_vp)]),
    ("glu_dist_prepare", _int, [_vp, _sz, _sz]),
    ("glu_dist_sort_begin", _int, [_vp, _vp, _vp, _sz, _vp, _P(_sz)]),
    ("glu_dist_sort_finish", _int, [_vp, _vp, _vp, _sz, _vp]),
    ("glu_dist_sort_ptr", _int, [_vp, _vp, _vp, _sz, _vp, _P(_vp), _P(_vp), _P(_sz)]),
    ("glu_dist_last_local_sort", _int, [_vp, _P(_u32)]),
    ("glu_dist_set_rounds", _int, [_vp, _int]),
    ("glu_dist_last_rounds", _int, [_vp, _P(_u32)]),
    ("glu_dist_set_reserved_cus", _int, [_vp, _int]),
    ("glu_dist_set_profiling", _int, [_vp, _int]),
    ("glu_dist_phase_times", _int, [_vp, _P(ctypes.c_double), _P(_u64)]),
    ("glu_dist_plan_buckets", _int, [_P(_u32), _int, _P(_int)]),
    ("glu_dist_plan_counts", _int, [_P(_u32), _int, _int, _P(_int), _P(_u64), _P(_u64)]),
    ("glu_dist_plan_groups", _int, [_P(_u32), _int, _P(_int), _int, _P(_int)]),
    ("glu_timer_begin", _int, [_P(_vp)]),
    ("glu_timer_end", _int, [_vp, _P(_u64)]),
]

_lib = None


class GluError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("glu_hip status %d: %s" % (status, message))
        self.status = status
        self.message = message


def lib():
    """Loads libglu_hip.so (built by __graft_entry__.build() / make -C gl-radix-sort_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != GLU_OK:
        raise GluError(status, lib().glu_last_error().decode())


def device_count():
    c = _int(0)
    check(lib().glu_device_count(ctypes.byref(c)))
    return c.value


def set_device(i):
    check(lib().glu_set_device(i))


def device_info():
    buf = ctypes.create_string_buffer(256)
    check(lib().glu_device_info(buf, 256))
    return buf.value.decode()


def synchronize():
    check(lib().glu_device_synchronize())


def queue():
    s = _vp()
    check(lib().glu_queue(ctypes.byref(s)))
    return s.value


class ShaderStorageBuffer:
    """Python mirror of glu::ShaderStorageBuffer (reference glu/gl_utils.hpp:146-246) over the C ABI."""

    def __init__(self, data=None, size=0):
        self._h = _u32(0)
        self._size = 0
        if data is not None:
            a = np.ascontiguousarray(data)
            if a.nbytes == 0:
                raise GluError(GLU_ERROR_INVALID_ARGUMENT, "empty data")
            check(lib().glu_buffer_create_with_data(a.ctypes.data_as(_vp), a.nbytes, ctypes.byref(self._h)))
            self._size = a.nbytes
        elif size > 0:
            self.resize(size)

    @classmethod
    def wrap(cls, device_ptr, size):
        b = cls()
        check(lib().glu_buffer_wrap(_vp(device_ptr), size, ctypes.byref(b._h)))
        b._size = size
        return b

    def handle(self):
        return self._h.value

    def size(self):
        return self._size

    def device_ptr(self):
        p = _vp()
        check(lib().glu_buffer_device_ptr(self._h, ctypes.byref(p)))
        return p.value

    def resize(self, size, keep_data=False):
        if size == self._size:
            return
        new = _u32(0)
        check(lib().glu_buffer_create(size, ctypes.byref(new)))
        if keep_data and self._h.value:
            check(lib().glu_buffer_copy(self._h, new, min(self._size, size), 0, 0))
        if self._h.value:
            check(lib().glu_buffer_destroy(self._h))
        self._h, self._size = new, size

    def clear(self, value=0):
        check(lib().glu_buffer_fill_u32(self._h, value))

    def write_data(self, data):
        a = np.ascontiguousarray(data)
        check(lib().glu_buffer_write(self._h, a.ctypes.data_as(_vp), a.nbytes, 0))

    def get_data(self, dtype):
        dt = np.dtype(dtype)
        if self._size % dt.itemsize:
            raise GluError(GLU_ERROR_INVALID_ARGUMENT, "Size %d isn't a multiple of %d" % (self._size, dt.itemsize))
        out = np.empty(self._size // dt.itemsize, dtype=dt)
        if self._size:
            check(lib().glu_buffer_read(self._h, out.ctypes.data_as(_vp), self._size, 0))
        return out

    def destroy(self):
        if self._h.value and _lib is not None:
            _lib.glu_buffer_destroy(self._h)
        self._h = _u32(0)
        self._size = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def plan_finish(count, key_bytes=4):
    """(first_capacity, last_capacity) of the in-LDS pass's tiles a whole-key sort of `count` elements enqueues by default; (0, 0):
    no attempt to end in LDS (glu_radix_sort_plan_finish; host only)."""
    a, b = _u32(0), _u32(0)
    check(lib().glu_radix_sort_plan_finish(count, key_bytes, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


class RadixSort:
    """glu::RadixSort (reference glu/RadixSort.hpp:186-354) over the C ABI."""

    def __init__(self, digit_bits=None, options=None):
        """options: {name: integer} switches for tests / tuning (glu_radix_sort_set_option; names as in the environment, with or
        without the GLU_HIP_ prefix) -- set on THIS object, the process environment is not touched."""
        self._h = _vp()
        check(lib().glu_radix_sort_create(ctypes.byref(self._h)))
        if digit_bits is not None:
            check(lib().glu_radix_sort_set_digit_bits(self._h, digit_bits))
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        check(lib().glu_radix_sort_set_option(self._h, str(name).encode(), int(value)))

    def set_digit_bits(self, bits):
        check(lib().glu_radix_sort_set_digit_bits(self._h, bits))

    @property
    def digit_bits(self):
        b = _u32(0)
        check(lib().glu_radix_sort_get_digit_bits(self._h, ctypes.byref(b)))
        return b.value

    def prepare_internal_buffers(self, count, key_bytes=4, with_vals=True):
        """Grow-only scratch for `count` elements (RadixSort.hpp:237-271); key_bytes 4 or 8, with_vals False for
        keys-only sorts.  After it the matching run call allocates nothing."""
        check(lib().glu_radix_sort_prepare_ex(self._h, count, key_bytes, 1 if with_vals else 0))

    def scratch_size(self):
        s = _sz(0)
        check(lib().glu_radix_sort_scratch_size(self._h, ctypes.byref(s)))
        return s.value

    def scratch_placement(self):
        """What the last prepare measured when it placed the large scratch arrays (glu_radix_sort_scratch_placement):
        {"candidates": n, "chosen_ms": t, "slowest_ms": t}; candidates == 0: no measurement was made."""
        n, a, b = _u32(0), ctypes.c_double(0), ctypes.c_double(0)
        check(lib().glu_radix_sort_scratch_placement(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b)))
        return {"candidates": int(n.value), "chosen_ms": float(a.value), "slowest_ms": float(b.value)}

    def set_profiling(self, enable):
        """True: events at every kernel boundary of a pass; "light": only around the kernel that moves a pass's data (the scatter,
        the in-LDS pass) -- a few microseconds of queue time per event saved; False: off."""
        check(lib().glu_radix_sort_set_profiling(self._h, 2 if enable == "light" else (1 if enable else 0)))

    def read_profile(self):
        """{count_ms, scan_ms, scatter_ms, passes}: summed device time per kernel class since the last read."""
        c, s, x = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        n = _u64(0)
        f, fn = ctypes.c_double(0), _u64(0)
        check(lib().glu_radix_sort_read_profile_finish(self._h, ctypes.byref(c), ctypes.byref(s), ctypes.byref(x),
                                                       ctypes.byref(n), ctypes.byref(f), ctypes.byref(fn)))
        return {"count_ms": c.value, "scan_ms": s.value, "scatter_ms": x.value, "passes": n.value,
                "finish_ms": f.value, "finish_passes": fn.value}

    def read_plan(self, passes, roles=False):
        """(skipped[passes], counted_alone[passes]) -- with roles=True also pair_role[passes] -- of the last planned sort
        (>= 2^22 elements); synchronise first."""
        sk, ca, ro = (_u32 * passes)(), (_u32 * passes)(), (_u32 * passes)()
        check(lib().glu_radix_sort_read_plan(self._h, sk, ca, ro, passes))
        return (list(sk), list(ca), list(ro)) if roles else (list(sk), list(ca))

    def read_finish(self):
        """{attempted, accepted, longest_run, capacity} of the last sort: did it try to / did it end in LDS
        (glu_radix_sort_read_finish); synchronise first."""
        a, b, c, d, e = _u32(0), _u32(0), _u32(0), _u32(0), _u32(0)
        check(lib().glu_radix_sort_read_finish(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d),
                                               ctypes.byref(e)))
        return {"attempted": a.value, "accepted": b.value, "longest_run": c.value, "capacity": d.value, "top_bit": e.value}

    def read_long_runs(self):
        """{runs, sub_blocks, pairs}: what the last sort that ended in LDS gave to the segmented passes for runs longer than the tile
        (glu_radix_sort_read_long_runs); waits for the device."""
        a, b, c = _u32(0), _u32(0), _u32(0)
        check(lib().glu_radix_sort_read_long_runs(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"runs": a.value, "sub_blocks": b.value, "pairs": c.value}

    def read_seg_finish(self):
        """{attempted, accepted, longest_run, capacity, runs, tile, split} of the last SEGMENTED sort: did it try to / did it end
        in LDS (glu_radix_sort_read_seg_finish); waits for the device."""
        a, b, c, d, e, f, g = (_u32(0) for _ in range(7))
        check(lib().glu_radix_sort_read_seg_finish(self._h, *(ctypes.byref(x) for x in (a, b, c, d, e, f, g))))
        return {"attempted": a.value, "accepted": b.value, "longest_run": c.value, "capacity": d.value, "runs": e.value,
                "tile": f.value, "split": g.value}

    def __call__(self, key_buffer, val_buffer, count, num_steps=0, key_bytes=4):
        kb = key_buffer.handle() if isinstance(key_buffer, ShaderStorageBuffer) else key_buffer
        vb = val_buffer.handle() if isinstance(val_buffer, ShaderStorageBuffer) else val_buffer
        fn = lib().glu_radix_sort_run if key_bytes == 4 else lib().glu_radix_sort_run_u64
        check(fn(self._h, kb, vb, count, num_steps))

    def sort_keys(self, key_buffer, count, num_steps=0):
        """Keys-only sort of a uint32 buffer (no value buffer at all)."""
        kb = key_buffer.handle() if isinstance(key_buffer, ShaderStorageBuffer) else key_buffer
        check(lib().glu_radix_sort_run_keys(self._h, kb, count, num_steps))

    def sort_keys_ptr(self, keys_ptr, count, num_steps=0, stream=None, key_bytes=4):
        fn = lib().glu_radix_sort_run_keys_ptr if key_bytes == 4 else lib().glu_radix_sort_run_keys_u64_ptr
        check(fn(self._h, _vp(keys_ptr), count, num_steps, _vp(stream)))

    KEY_TYPES = {"uint32": 0, "int32": 1, "float32": 2, "uint64": 3, "int64": 4, "float64": 5}

    def sort_typed_ptr(self, keys_ptr, vals_ptr, count, key_type, stream=None):
        """key_type: a numpy dtype name in KEY_TYPES; vals_ptr may be None (keys only)."""
        check(lib().glu_radix_sort_run_typed_ptr(self._h, _vp(keys_ptr), _vp(vals_ptr), count, self.KEY_TYPES[key_type],
                                                 _vp(stream)))

    def sort_bit_range_ptr(self, keys_ptr, vals_ptr, count, begin_bit, end_bit, stream=None, key_bytes=4):
        """Stable sort by the key bits [begin_bit, end_bit) only; vals_ptr may be None (keys only)."""
        check(lib().glu_radix_sort_run_bit_range_ptr(self._h, _vp(keys_ptr), _vp(vals_ptr), count, key_bytes * 8, begin_bit,
                                                     end_bit, _vp(stream)))

    def run_ptr(self, keys_ptr, vals_ptr, count, num_steps=0, stream=None, key_bytes=4):
        fn = lib().glu_radix_sort_run_ptr if key_bytes == 4 else lib().glu_radix_sort_run_u64_ptr
        check(fn(self._h, _vp(keys_ptr), _vp(vals_ptr), count, num_steps, _vp(stream)))

    def run_segments_ptr(self, in_keys, in_vals, out_keys, out_vals, count, piece_begin, piece_len, piece_segment,
                         num_segments, key_bits, stream=None):
        """Segmented stable sort (glu_radix_sort_run_segments_ptr): the pieces [piece_begin[i], + piece_len[i]) of the
        input arrays, grouped into segments piece_segment[i], leave as segments in ascending order, each stably sorted by
        its low key_bits bits; the input arrays are clobbered."""
        pb = np.ascontiguousarray(piece_begin, dtype=np.uint64)
        pl = np.ascontiguousarray(piece_len, dtype=np.uint64)
        ps = np.ascontiguousarray(piece_segment, dtype=np.uint32)
        assert pb.shape == pl.shape == ps.shape and pb.ndim == 1
        check(lib().glu_radix_sort_run_segments_ptr(
            self._h, _vp(in_keys), _vp(in_vals), _vp(out_keys), _vp(out_vals), count,
            pb.ctypes.data_as(_P(_u64)), pl.ctypes.data_as(_P(_u64)), ps.ctypes.data_as(_P(_u32)), pb.shape[0],
            num_segments, key_bits, _vp(stream)))

    def partition_ptr(self, src_keys, src_vals, dst_keys, dst_vals, count, shift, bits, histogram_ptr=None,
                      stream=None):
        check(lib().glu_radix_sort_partition_ptr(self._h, _vp(src_keys), _vp(src_vals), _vp(dst_keys), _vp(dst_vals),
                                                 count, shift, bits, _vp(histogram_ptr), _vp(stream)))

    def destroy(self):
        if self._h and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.glu_radix_sort_destroy(self._h)
        self._h = _vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class BlellochScan:
    """glu::BlellochScan (reference glu/BlellochScan.hpp:80-191) over the C ABI."""

    def __init__(self, data_type):
        self._h = _vp()
        check(lib().glu_scan_create(data_type, ctypes.byref(self._h)))

    def __call__(self, buffer, count, num_partitions=1):
        b = buffer.handle() if isinstance(buffer, ShaderStorageBuffer) else buffer
        check(lib().glu_scan_run(self._h, b, count, num_partitions))

    def prepare(self, count, num_partitions=1):
        """Scratch for scans of `count` x `num_partitions` elements: after it a scan allocates nothing (capturable)."""
        check(lib().glu_scan_prepare(self._h, count, num_partitions))

    def run_ptr(self, data_ptr, count, num_partitions=1, stream=None):
        check(lib().glu_scan_run_ptr(self._h, _vp(data_ptr), count, num_partitions, _vp(stream)))

    def destroy(self):
        if self._h and _lib is not None:
            _lib.glu_scan_destroy(self._h)
        self._h = _vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Reduce:
    """glu::Reduce (reference glu/Reduce.hpp:51-136) over the C ABI."""

    def __init__(self, data_type, operator_):
        self._h = _vp()
        check(lib().glu_reduce_create(data_type, operator_, ctypes.byref(self._h)))

    def __call__(self, buffer, count):
        b = buffer.handle() if isinstance(buffer, ShaderStorageBuffer) else buffer
        check(lib().glu_reduce_run(self._h, b, count))

    def run_ptr(self, data_ptr, count, stream=None):
        check(lib().glu_reduce_run_ptr(self._h, _vp(data_ptr), count, _vp(stream)))

    def destroy(self):
        if self._h and _lib is not None:
            _lib.glu_reduce_destroy(self._h)
        self._h = _vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def measure_elapsed_time(callback):
    """glu::measure_gl_elapsed_time (reference glu/gl_utils.hpp:249-265): nanoseconds of device time."""
    t = _vp()
    check(lib().glu_timer_begin(ctypes.byref(t)))
    callback()
    ns = _u64(0)
    check(lib().glu_timer_end(t, ctypes.byref(ns)))
    return ns.value


DIST_UNIQUE_ID_BYTES = 128
DIST_BUCKETS = 256


def plan_segments(piece_begin, piece_len, piece_segment, num_segments, num_workgroups):
    """glu_radix_sort_plan_segments (host only): -> (sub_blocks [n][2], workgroup_first, segment_first, segment_start)."""
    pb = np.ascontiguousarray(piece_begin, dtype=np.uint64)
    pl = np.ascontiguousarray(piece_len, dtype=np.uint64)
    ps = np.ascontiguousarray(piece_segment, dtype=np.uint32)
    cap = pb.shape[0] + num_workgroups
    subs = np.zeros((cap, 2), dtype=np.uint32)
    wg = np.zeros(num_workgroups + 1, dtype=np.uint32)
    sf = np.zeros(num_segments + 1, dtype=np.uint32)
    ss = np.zeros(num_segments + 1, dtype=np.uint64)
    n = _sz(0)
    check(lib().glu_radix_sort_plan_segments(pb.ctypes.data_as(_P(_u64)), pl.ctypes.data_as(_P(_u64)), ps.ctypes.data_as(_P(_u32)),
                                             pb.shape[0], num_segments, num_workgroups, subs.ctypes.data_as(_P(_u32)), cap,
                                             wg.ctypes.data_as(_P(_u32)), sf.ctypes.data_as(_P(_u32)), ss.ctypes.data_as(_P(_u64)),
                                             ctypes.byref(n)))
    return subs[:n.value], wg, sf, ss


def dist_available():
    """glu_dist_available: raises GluError unless this process can load RCCL with the entry points glu_dist_* needs."""
    check(lib().glu_dist_available())


def dist_unique_id():
    """Rank 0: the RCCL unique id (bytes) to hand to every rank's Dist(...)."""
    buf = ctypes.create_string_buffer(DIST_UNIQUE_ID_BYTES)
    check(lib().glu_dist_unique_id(buf, DIST_UNIQUE_ID_BYTES))
    return buf.raw


def dist_plan(all_hist, world_size, rank):
    """The C ABI's plan (host only): (bucket_owner[256], send_counts[world], recv_counts[world]) as Python lists."""
    flat = [int(x) for row in all_hist for x in row]
    assert len(flat) == world_size * DIST_BUCKETS
    h = (_u32 * len(flat))(*flat)
    owner = (_int * DIST_BUCKETS)()
    check(lib().glu_dist_plan_buckets(h, world_size, owner))
    send, recv = (_u64 * world_size)(), (_u64 * world_size)()
    check(lib().glu_dist_plan_counts(h, world_size, rank, owner, send, recv))
    return list(owner), list(send), list(recv)


def dist_plan_groups(all_hist, world_size, owner, rounds):
    """glu_dist_plan_groups (host only): [world][rounds + 1] bucket boundaries of every rank's groups."""
    flat = [int(x) for row in all_hist for x in row]
    h = (_u32 * len(flat))(*flat)
    own = (_int * DIST_BUCKETS)(*[int(x) for x in owner])
    cut = (_int * (world_size * (rounds + 1)))()
    check(lib().glu_dist_plan_groups(h, world_size, own, rounds, cut))
    return [list(cut[q * (rounds + 1):(q + 1) * (rounds + 1)]) for q in range(world_size)]


class Dist:
    """glu_dist over the C ABI: the sharded sort of one rank (RCCL communicator + local sorter inside the library)."""

    def __init__(self, unique_id, world_size, rank):
        self._h = _vp()
        check(lib().glu_dist_create(unique_id, len(unique_id), world_size, rank, ctypes.byref(self._h)))
        self.world_size, self.rank = world_size, rank

    def prepare(self, local_count, recv_capacity=0):
        check(lib().glu_dist_prepare(self._h, local_count, recv_capacity))

    @property
    def sorter(self):
        """The local RadixSort inside the object (borrowed: profiling / digit width, never destroyed from here)."""
        h = _vp()
        check(lib().glu_dist_local_sorter(self._h, ctypes.byref(h)))
        s = RadixSort.__new__(RadixSort)
        s._h = h
        s._borrowed = True
        return s

    def sort_begin(self, keys_ptr, vals_ptr, local_count, stream=None):
        n = _sz(0)
        check(lib().glu_dist_sort_begin(self._h, _vp(keys_ptr), _vp(vals_ptr), local_count, _vp(stream), ctypes.byref(n)))
        return n.value

    def sort_finish(self, recv_keys_ptr, recv_vals_ptr, capacity, stream=None):
        check(lib().glu_dist_sort_finish(self._h, _vp(recv_keys_ptr), _vp(recv_vals_ptr), capacity, _vp(stream)))

    def sort_ptr(self, keys_ptr, vals_ptr, local_count, stream=None):
        """-> (device pointer of the shard's keys, of its values, count); the arrays belong to the object."""
        k, v, n = _vp(), _vp(), _sz(0)
        check(lib().glu_dist_sort_ptr(self._h, _vp(keys_ptr), _vp(vals_ptr), local_count, _vp(stream), ctypes.byref(k),
                                      ctypes.byref(v), ctypes.byref(n)))
        return k.value or 0, v.value or 0, n.value

    def partition_shift(self):
        sh = _u32(0)
        check(lib().glu_dist_partition_shift(self._h, ctypes.byref(sh)))
        return sh.value

    def last_local_sort(self):
        """'segmented' (three passes over the low 24 bits per bucket) or 'ordinary' (all 32 bits): the last sort's local sort."""
        v = _u32(0)
        check(lib().glu_dist_last_local_sort(self._h, ctypes.byref(v)))
        return "segmented" if v.value else "ordinary"

    def set_rounds(self, rounds):
        """Rounds of the exchange (glu_dist_set_rounds; the same value on every rank)."""
        check(lib().glu_dist_set_rounds(self._h, int(rounds)))

    def last_rounds(self):
        """In how many rounds the last sort's exchange was posted (glu_dist_last_rounds)."""
        v = _u32(0)
        check(lib().glu_dist_last_rounds(self._h, ctypes.byref(v)))
        return int(v.value)

    def set_reserved_cus(self, cus):
        check(lib().glu_dist_set_reserved_cus(self._h, cus))

    def set_profiling(self, enable):
        check(lib().glu_dist_set_profiling(self._h, 1 if enable else 0))

    def phase_times(self):
        ms = (ctypes.c_double * 4)()
        n = _u64(0)
        check(lib().glu_dist_phase_times(self._h, ms, ctypes.byref(n)))
        names = ("partition", "histogram_exchange_and_plan", "all_to_all", "local_sort")
        return {"sorts": int(n.value), **{k: float(ms[i]) for i, k in enumerate(names)}}

    def destroy(self):
        if self._h and _lib is not None:
            _lib.glu_dist_destroy(self._h)
        self._h = _vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
