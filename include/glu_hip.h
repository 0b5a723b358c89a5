/*
 * glu_hip.h -- C ABI of libglu_hip.so: MI355X (gfx950) radix sort / exclusive scan / reduce.
 *
 * This is the drop-in boundary for the hot path of loryruta/gl-radix-sort (reference @ v2).  The reference
 * has no FFI layer: its operators are header-only C++ classes (namespace glu) that record OpenGL compute
 * dispatches against GLuint buffer names.  The functions below are what those classes need from the device
 * side; the C++17 headers in gl-radix-sort_amd/glu/ re-create the reference's classes on top of them, and any
 * other host language binds the same symbols (INTEGRATION.md shows the bindings).
 *
 * Conventions
 *   - every function returns glu_status (0 = GLU_OK); nothing here prints or calls exit() -- the
 *     print-and-exit(1) convention of glu/errors.hpp:8-18 lives in the C++ headers above this ABI;
 *   - glu_last_error() returns a thread-local message for the last non-zero status;
 *   - buffers are named by 32-bit handles exactly like GLuint buffer names: 0 is "no buffer"
 *     (checked the way glu/RadixSort.hpp:275-276 checks it);
 *   - all handle-based calls enqueue on one in-order queue per process (a HIP stream owned by the
 *     library, the analogue of the thread's GL command queue) and return without waiting for the GPU,
 *     like the reference's operator() (glu/RadixSort.hpp:273-334 never blocks).  glu_buffer_read() and
 *     glu_device_synchronize() wait.  *_ptr entry points take raw device pointers plus a caller stream
 *     (a hipStream_t passed as void*; NULL = the library queue -- to target HIP's null stream pass the
 *     hipStreamLegacy / hipStreamPerThread handle, not 0);
 *   - no function allocates device memory inside a sort/scan/reduce call once the matching
 *     *_prepare() has been called with a count at least as large (glu/RadixSort.hpp:237-271:
 *     grow-only scratch).
 *   - not thread-safe per handle; distinct handles may be used from distinct host threads (every entry point makes the
 *     library's device the calling thread's current HIP device; per-kernel one-time setup is guarded).  The
 *     *_ptr entry points accept NULL arrays when count == 0 (an empty shard).
 */
#ifndef GLU_HIP_H
#define GLU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(GLU_HIP_BUILD)
#define GLU_API __attribute__((visibility("default")))
#else
#define GLU_API
#endif

typedef int glu_status;
enum
{
    GLU_OK = 0,
    GLU_ERROR_INVALID_ARGUMENT = 1, /* what GLU_CHECK_ARGUMENT rejects in the reference */
    GLU_ERROR_INVALID_STATE = 2,    /* what GLU_CHECK_STATE rejects */
    GLU_ERROR_OUT_OF_MEMORY = 3,
    GLU_ERROR_DEVICE = 4,           /* a HIP runtime call failed; message has the hipError string */
    GLU_ERROR_NO_DEVICE = 5         /* no gfx950 GPU visible: there is NO CPU fallback */
};

/* glu/data_types.hpp:8-22 -- same numeric values */
typedef enum glu_data_type
{
    GLU_DATA_TYPE_FLOAT = 0,
    GLU_DATA_TYPE_DOUBLE,
    GLU_DATA_TYPE_INT,
    GLU_DATA_TYPE_UINT,
    GLU_DATA_TYPE_VEC2,
    GLU_DATA_TYPE_VEC4,
    GLU_DATA_TYPE_DVEC2,
    GLU_DATA_TYPE_DVEC4,
    GLU_DATA_TYPE_UVEC2,
    GLU_DATA_TYPE_UVEC4,
    GLU_DATA_TYPE_IVEC2,
    GLU_DATA_TYPE_IVEC4,
    GLU_DATA_TYPE_COUNT_
} glu_data_type;

/* glu/Reduce.hpp:42-48 -- same numeric values */
typedef enum glu_reduce_operator
{
    GLU_REDUCE_SUM = 0,
    GLU_REDUCE_MUL,
    GLU_REDUCE_MIN,
    GLU_REDUCE_MAX,
    GLU_REDUCE_COUNT_
} glu_reduce_operator;

typedef unsigned int glu_buffer; /* replaces GLuint buffer names (glu/gl_utils.hpp:149) */
typedef struct glu_radix_sort_s* glu_radix_sort;
typedef struct glu_scan_s* glu_scan;
typedef struct glu_reduce_s* glu_reduce;
typedef struct glu_timer_s* glu_timer;
typedef struct glu_dist_s* glu_dist;

/* ---- library / device ---------------------------------------------------------------------------- */

/* Thread-local text for the last failing call on this thread ("" if none). */
GLU_API const char* glu_last_error(void);
/* "glu_hip <version> gfx950" */
GLU_API const char* glu_version(void);
/* Number of visible HIP devices (0 is not an error here). */
GLU_API glu_status glu_device_count(int* count);
/* Select the device the library queue lives on (default: the current HIP device, normally 0).  Must be
 * called before the first buffer / operator is created; one device per process (one process per GPU). */
GLU_API glu_status glu_set_device(int device);
/* Human readable device line: name, gcnArchName, CU count, memory. */
GLU_API glu_status glu_device_info(char* out, size_t out_size);
/* Block until everything enqueued on the library queue has finished (the glFinish analogue). */
GLU_API glu_status glu_device_synchronize(void);
/* The library queue as a hipStream_t (void*), for callers that want to enqueue their own work in order. */
GLU_API glu_status glu_queue(void** stream);

/* ---- buffers: replaces glu::ShaderStorageBuffer's GL calls (glu/gl_utils.hpp:146-246) ------------- */

/* glCreateBuffers + glBufferStorage(size, NULL)      (gl_utils.hpp:203-205).  size may be 0. */
GLU_API glu_status glu_buffer_create(size_t size, glu_buffer* out);
/* glBufferStorage(size, data)                          (gl_utils.hpp:165-167) */
GLU_API glu_status glu_buffer_create_with_data(const void* data, size_t size, glu_buffer* out);
/* Name an existing device allocation (e.g. a torch tensor's data_ptr()); not owned, never freed. */
GLU_API glu_status glu_buffer_wrap(void* device_ptr, size_t size, glu_buffer* out);
/* glDeleteBuffers                                      (gl_utils.hpp:186-187).  0 is ignored. */
GLU_API glu_status glu_buffer_destroy(glu_buffer buffer);
GLU_API glu_status glu_buffer_size(glu_buffer buffer, size_t* size);
GLU_API glu_status glu_buffer_device_ptr(glu_buffer buffer, void** device_ptr);
/* glBufferSubData(offset, size, data)                  (gl_utils.hpp:221-227) */
GLU_API glu_status glu_buffer_write(glu_buffer buffer, const void* data, size_t size, size_t offset);
/* glGetBufferSubData(offset, size, data); waits        (gl_utils.hpp:229-238) */
GLU_API glu_status glu_buffer_read(glu_buffer buffer, void* data, size_t size, size_t offset);
/* glClearBufferData(GL_R32UI, value): whole buffer     (gl_utils.hpp:215-219) */
GLU_API glu_status glu_buffer_fill_u32(glu_buffer buffer, uint32_t value);
/* glCopyBufferSubData                                  (gl_utils.hpp:13-22) */
GLU_API glu_status glu_buffer_copy(glu_buffer src, glu_buffer dst, size_t size, size_t src_offset, size_t dst_offset);

/* ---- radix sort: replaces glu::RadixSort (glu/RadixSort.hpp:186-354) ------------------------------ */

/* RadixSort::RadixSort()                               (RadixSort.hpp:205-233) */
GLU_API glu_status glu_radix_sort_create(glu_radix_sort* out);
GLU_API glu_status glu_radix_sort_destroy(glu_radix_sort sort);
/* RadixSort::prepare_internal_buffers(count)           (RadixSort.hpp:237-271): grow-only scratch for
 * `count` pairs with 32-bit keys.
 * Cost: for less than 512 MiB of keys, the allocations.  From there on (2^27 32-bit keys) prepare also PLACES the key and
 * value scratch by measurement (glu_radix_sort_scratch_placement below): 0.13-1 s of host time, usually 0.3 s, up to 48
 * calibration sorts on the library queue behind a hipDeviceSynchronize, and transiently about 2 x (4 arrays of the prepared
 * size + 7.5 GiB of spacers) of device memory.  When hipMemGetInfo reports less free memory than that the search is skipped
 * (GLU_VERBOSE says so) and the arrays are two plain hipMallocs: everything works, large sorts run 5-8 % slower on some
 * placements.  Only the glu_radix_sort_prepare* / glu_dist_prepare calls do this: a sort on an object that was not prepared
 * for its size grows the scratch with plain allocations (hipFree + hipMalloc: a device-wide synchronisation, not
 * capturable) and never measures. */
GLU_API glu_status glu_radix_sort_prepare(glu_radix_sort sort, size_t count);
/* Same for 64-bit keys (BASELINE.json config 5). */
GLU_API glu_status glu_radix_sort_prepare_u64(glu_radix_sort sort, size_t count);
/* The general form (RadixSort.hpp:237-271 knows one key type only): scratch for `count` elements of `key_bytes`-byte
 * keys (4 or 8), with or without a value array.  After it, the matching run entry point (run / run_u64 / run_keys* /
 * run_typed / run_bit_range with that key width) allocates nothing for counts up to `count`.  A run of a wider key type
 * than prepared for still works but grows the scratch inside the call (hipFree + hipMalloc: a device-wide
 * synchronisation, and not capturable into a graph). */
GLU_API glu_status glu_radix_sort_prepare_ex(glu_radix_sort sort, size_t count, size_t key_bytes, int with_vals);
/* RadixSort::operator()(key_buffer, val_buffer, count, num_steps)   (RadixSort.hpp:273-334).
 * Stable ascending sort of `count` (uint32 key, uint32 val) pairs by the low 4*num_steps key bits
 * (num_steps == 0 or > 8: all 32 bits).  count <= 1 returns immediately (:278).  count must be < 2^32.
 * Deliberate deviation: the result is always left in key_buffer / val_buffer; the
 * reference leaves it in its private scratch buffers when num_steps is odd.
 * Every sort entry point below behaves the same way: the call only enqueues work (no host synchronisation, no allocation
 * once prepared: it can be captured into a HIP graph); up to 16384 pairs it is one kernel launch; from 2^22 elements up a
 * counting pass whose digit is the same in every key is skipped on the device (same result, less work). */
GLU_API glu_status glu_radix_sort_run(glu_radix_sort sort, glu_buffer key_buffer, glu_buffer val_buffer, size_t count,
                                      size_t num_steps);
/* Raw-pointer form of the same call; `stream` is a hipStream_t or NULL. */
GLU_API glu_status glu_radix_sort_run_ptr(glu_radix_sort sort, uint32_t* keys, uint32_t* vals, size_t count,
                                          size_t num_steps, void* stream);
/* 64-bit keys + 32-bit payload, num_steps 4-bit digits (0 or > 16: all 64 bits). */
GLU_API glu_status glu_radix_sort_run_u64(glu_radix_sort sort, glu_buffer key_buffer, glu_buffer val_buffer,
                                          size_t count, size_t num_steps);
GLU_API glu_status glu_radix_sort_run_u64_ptr(glu_radix_sort sort, uint64_t* keys, uint32_t* vals, size_t count,
                                              size_t num_steps, void* stream);
/* Keys-only sorts (not in the reference, whose value buffer is mandatory -- README.md:88-89 tells users to allocate a
 * dummy one): same ordering of the keys, no value traffic (12 instead of 20 bytes per key and pass). */
GLU_API glu_status glu_radix_sort_run_keys(glu_radix_sort sort, glu_buffer key_buffer, size_t count, size_t num_steps);
GLU_API glu_status glu_radix_sort_run_keys_ptr(glu_radix_sort sort, uint32_t* keys, size_t count, size_t num_steps,
                                               void* stream);
GLU_API glu_status glu_radix_sort_run_keys_u64_ptr(glu_radix_sort sort, uint64_t* keys, size_t count, size_t num_steps,
                                                   void* stream);
/* Typed keys (not in the reference: uint32 only).  Signed integers and IEEE floats are sorted in their natural order
 * (floats as a total order: -0 < +0, NaNs beyond the infinities of their sign) by encoding the key on the first pass's
 * loads and decoding on the last pass's stores -- no extra pass over memory.  vals may be NULL (keys only). */
typedef enum glu_key_type
{
    GLU_KEY_UINT32 = 0,
    GLU_KEY_INT32,
    GLU_KEY_FLOAT32,
    GLU_KEY_UINT64,
    GLU_KEY_INT64,
    GLU_KEY_FLOAT64
} glu_key_type;
GLU_API glu_status glu_radix_sort_run_typed_ptr(glu_radix_sort sort, void* keys, uint32_t* vals, size_t count,
                                                glu_key_type key_type, void* stream);
/* Stable sort by the key bits [begin_bit, end_bit) only (not in the reference, whose num_steps always starts at bit 0,
 * RadixSort.hpp:303,331-332): keys that are known to fit 24 bits sort in 3 passes instead of 4, a sort by the high bits
 * alone leaves the low-bit order untouched.  key_bits is 32 or 64 (unsigned keys), vals may be NULL (keys only), the
 * result is in the caller's arrays for every pass count; begin_bit == end_bit is a no-op. */
GLU_API glu_status glu_radix_sort_run_bit_range_ptr(glu_radix_sort sort, void* keys, uint32_t* vals, size_t count,
                                                    uint32_t key_bits, uint32_t begin_bit, uint32_t end_bit, void* stream);
/* One stable counting pass on the digit (key >> shift) & ((1 << bits) - 1), 1 <= bits <= 8, from src to dst
 * (distinct buffers).  This is the partition step of the multi-GPU sort (top-8-bit buckets).  If
 * digit_histogram != NULL it receives the 1 << bits digit totals (device memory, uint32). */
GLU_API glu_status glu_radix_sort_partition_ptr(glu_radix_sort sort, const uint32_t* src_keys, const uint32_t* src_vals,
                                                uint32_t* dst_keys, uint32_t* dst_vals, size_t count, uint32_t shift,
                                                uint32_t bits, uint32_t* digit_histogram, void* stream);
/* Segmented stable sort (not in the reference; it is the local sort of the sharded sort below, where a rank's shard
 * arrives as one message per source rank, each grouped by top-byte bucket): the input arrays hold `num_pieces` pieces,
 * piece i = elements [piece_begin[i], piece_begin[i] + piece_len[i]) of in_keys / in_vals, together exactly `count`
 * elements and no element twice (the pieces tile [0, count): GLU_ERROR_INVALID_ARGUMENT otherwise); piece i belongs to
 * segment piece_segment[i] < num_segments <= 2^24.  Segment g = its pieces laid end to end in the
 * order they are listed.  The output arrays receive the segments in ascending order of g, each stably sorted by the low
 * `key_bits` key bits (0, 8, 16, 24 or 32; 0 = only the regrouping).  Each pass is the reference's stable counting pass
 * (RadixSort.hpp:142-182) applied per segment, all segments in one launch sequence of the sort's own kernels; the first
 * pass reads the pieces where they lie, so the regrouping costs no pass of its own.  in != out; the input arrays are used
 * as scratch (their contents are lost).  Enqueues on `stream` (the piece arrays are read before the call returns); after
 * glu_radix_sort_prepare(sort, count) it allocates nothing on the device.  Not capturable into a graph (the sub-block
 * descriptors of a call are staged through pinned buffers that later calls reuse): GLU_ERROR_INVALID_STATE under capture. */
GLU_API glu_status glu_radix_sort_run_segments_ptr(glu_radix_sort sort, uint32_t* in_keys, uint32_t* in_vals,
                                                   uint32_t* out_keys, uint32_t* out_vals, size_t count,
                                                   const uint64_t* piece_begin, const uint64_t* piece_len,
                                                   const uint32_t* piece_segment, size_t num_pieces, uint32_t num_segments,
                                                   uint32_t key_bits, void* stream);
/* The host half of glu_radix_sort_run_segments_ptr as a pure function (no device needed: unit-testable): how a segmented pass
 * over these pieces is cut into sub-blocks for `num_workgroups` workgroups.  A sub-block is the part of one piece that falls
 * into one workgroup's equal share of the elements; sub-blocks are numbered segment-major, in the order of each segment's
 * elements.  sub_blocks: [*num_sub_blocks][2] = (begin, end) element range in the input arrays (room for sub_block_capacity
 * pairs; at most num_pieces + num_workgroups are needed); workgroup_first: [num_workgroups + 1], workgroup w runs the
 * sub-blocks [workgroup_first[w], workgroup_first[w + 1]); segment_first: [num_segments + 1] first sub-block of every
 * segment; segment_start: [num_segments + 1] where every segment starts in the output.  Any output array may be NULL. */
GLU_API glu_status glu_radix_sort_plan_segments(const uint64_t* piece_begin, const uint64_t* piece_len,
                                                const uint32_t* piece_segment, size_t num_pieces, uint32_t num_segments,
                                                uint32_t num_workgroups, uint32_t* sub_blocks, size_t sub_block_capacity,
                                                uint32_t* workgroup_first, uint32_t* segment_first, uint64_t* segment_start,
                                                size_t* num_sub_blocks);
/* Digit width (bits per counting pass) the sort uses internally: 4 (the reference's pass structure: 8 passes
 * for 32-bit keys) or 8 (4 passes).  The sorted result is identical; see DESIGN.md. */
GLU_API glu_status glu_radix_sort_set_digit_bits(glu_radix_sort sort, uint32_t bits);
GLU_API glu_status glu_radix_sort_get_digit_bits(glu_radix_sort sort, uint32_t* bits);
/* Switches of a sort object for tests, tuning and A/B runs (nothing the reference has: its only knob is num_steps,
 * RadixSort.hpp:273): `name` is one of the options listed in gl-radix-sort_amd/csrc/glu_hip.hip (kSortOptions: SORT_LDS_FINISH,
 * SORT_FINISH_MIN, SORT_PAIR_MIN, SORT_FORK, SORT_NO_LINES, ...), case-insensitive, with or without the prefix GLU_HIP_.  Takes effect from the
 * object's next prepare / sort.  The process environment (GLU_HIP_<NAME>=value) supplies DEFAULTS, read when an object is
 * created; a program never needs to touch it.  GLU_ERROR_INVALID_ARGUMENT for an unknown name. */
GLU_API glu_status glu_radix_sort_set_option(glu_radix_sort sort, const char* name, long long value);
/* Where the two large scratch arrays lie to each other decides between discrete speeds of every pass on this memory system,
 * so glu_radix_sort_prepare* places key + value scratch of 512 MiB of keys or more BY MEASUREMENT: the value array is
 * allocated behind spacers of 0, 0.5 .. 7.5 GiB (8 to 16 candidates; the search ends when one of them is 7 % faster
 * than the slowest seen, or after a second), each candidate sorts pseudo-random pairs of the prepared count
 * three times on the library queue, the fastest pair of arrays is kept, everything else is freed again (0.13-1 s, usually 0.3 s, once,
 * inside an explicit prepare call only -- no sort, no glu_dist_sort_* call measures or makes transient allocations;
 * GLU_HIP_SCRATCH_TUNE=0 takes the first allocation as the reference's prepare_internal_buffers does,
 * RadixSort.hpp:237-271).  This reports what the last such measurement saw: the number of candidates (0 = none was made:
 * switched off, too little free memory, arrays below 512 MiB, or the scratch was grown by a sort and not by prepare), the
 * calibration sort time of the chosen and of the slowest one. */
GLU_API glu_status glu_radix_sort_scratch_placement(glu_radix_sort sort, uint32_t* candidates, double* chosen_ms, double* slowest_ms);
/* Bytes of scratch currently owned by the sort object (keys + vals + tables). */
GLU_API glu_status glu_radix_sort_scratch_size(glu_radix_sort sort, size_t* bytes);
/* Per-kernel device timing (the measure_gl_elapsed_time idea, glu/gl_utils.hpp:249-265, at kernel granularity):
 * while enabled, every counting pass records HIP events on its stream around the count, scan and scatter
 * kernels.  glu_radix_sort_read_profile waits for the recorded work, returns the summed device milliseconds
 * per kernel class and the number of passes since the previous read, and resets the accumulation. */
/* enable = 2 ("light"): only the two events around the kernel that moves the data of a pass that is expected to run are
 * recorded (the scatter kernel of a counting pass, the in-LDS pass); count_ms / scan_ms then read 0.  An event between two
 * kernels costs the queue a few microseconds: the 28 of a large sort that ends in LDS were 4 % of its time. */
GLU_API glu_status glu_radix_sort_set_profiling(glu_radix_sort sort, int enable);
GLU_API glu_status glu_radix_sort_read_profile(glu_radix_sort sort, double* count_ms, double* scan_ms,
                                               double* scatter_ms, uint64_t* passes);
/* The same with the in-LDS pass of sorts that ended in LDS (glu_radix_sort_read_finish below): finish_ms = its summed
 * device milliseconds, finish_passes = how many ran.  Such a sort enqueues two sequences of passes of which one returns at
 * once; both calls book the passes that did the work (which sequence that was is read from the last sort and assumed for
 * every sort since the previous read): the two top-bit passes when the sort ended in LDS, else the ordinary passes plus the
 * count kernel the refused attempt cost. */
GLU_API glu_status glu_radix_sort_read_profile_finish(glu_radix_sort sort, double* count_ms, double* scan_ms,
                                                      double* scatter_ms, uint64_t* passes, double* finish_ms,
                                                      uint64_t* finish_passes);
/* Diagnostics of the last sort of >= 2^22 elements on this object, whose passes are planned on the device (the caller has
 * synchronised the sort's stream): for pass p < passes, skipped[p] != 0 if the pass was an identity (every key had the same
 * digit value: its scatter did not run) -- 1 if its count kernel found that out, 2 if it was known before counting (the
 * first pass of a sort of unsigned keys notes which key bits vary at all; a later pass on bits that do not vary reads
 * nothing); pair_role[p] = 1 / 2 if the pass was the first / second of a pair of passes that
 * share one read of the keys (large sorts with 8-bit digits: the first pass's count kernel also builds a two-digit
 * histogram, the second pass takes its count table from it), 0 if it stood alone; counted_alone[p] = 1 if a second pass
 * of a pair counted for itself after all (skewed digit values, a 16-bit counter overflow).  Any array may be NULL;
 * passes <= 32. */
GLU_API glu_status glu_radix_sort_read_plan(glu_radix_sort sort, uint32_t* skipped, uint32_t* counted_alone,
                                            uint32_t* pair_role, size_t passes);
/* Host only (no device needed): the tiles of the in-LDS pass a whole-key sort of `count` elements of 4- or 8-byte keys
 * enqueues by default -- first_capacity: the tile that suits uniformly drawn keys, last_capacity: the largest one enqueued
 * behind it; both 0: such a sort makes no attempt (too small, or its runs would outgrow the largest tile). */
GLU_API glu_status glu_radix_sort_plan_finish(size_t count, uint32_t key_bytes, uint32_t* first_capacity,
                                              uint32_t* last_capacity);
/* A sort that ends in LDS (replaces six of the reference's eight 4-bit steps, glu/RadixSort.hpp:289-333, by one pass).  A sort of
 * whole 32-bit or 64-bit keys (any key type) of 7 * 2^22 (keys only: 2^25; 64-bit keys: 3 * 2^21) .. about 2^29 elements with 8-bit digits first tries a
 * shorter way to the same result: two counting passes on 16 TOP key bits, after which the array is 65536 runs of keys that share
 * those bits, and one pass in which a workgroup per run orders the run by the remaining low bits inside LDS, in place -- 52.25
 * instead of 72.5 bytes of memory traffic per pair with 32-bit keys, 80.25 instead of 225 with 64-bit keys (which rank key bits
 * [32, 48) and repair ties exactly).  The device decides from exact run lengths before anything is moved: the in-LDS pass is
 * enqueued in the tile geometry that suits uniformly drawn keys of this count (1536 / 2560 / 4608 / 9216 pairs) and in the next
 * larger ones, and the device runs the smallest that holds the runs (`capacity`); runs longer than the tile go to segmented
 * passes (glu_radix_sort_read_long_runs below) or -- 64-bit, typed and keys-only sorts; keys crowded into few runs -- send the sort
 * to the ordinary passes, at the cost of one extra read of the keys (every sort asks: its cost does not depend on what the
 * object sorted before).  The launch sequence is the same either way (the kernels of the sequence not taken return at once), so the
 * sort stays asynchronous and capturable.
 * This reports, for the last sort on the object (the caller has synchronised its stream): attempted = 1 if both sequences were
 * enqueued, accepted = 1 if the sort ended in LDS, longest_run = the longest run counted (0xFFFFFFFF: not counted), capacity =
 * the tile the device chose (refused: the largest one enqueued).  glu_radix_sort_read_plan describes the ordinary passes (all
 * "skipped without counting" when accepted = 1).
 * top_bit: the runs were the values of key bits [top_bit - 16, top_bit).  Untyped keys: the device chooses it from a sample of
 * the keys -- the highest bit that varies in the sample + 1, at least 16 (64-bit keys: moved up to 40 or 48 where a digit would
 * straddle the key words) -- so keys below 2^28 make 65536 runs of bits [12, 28) on an object's FIRST sort; the exact bits
 * collected by the first count kernel check the sample: a missed bit refuses the attempt and becomes the floor of the next
 * sort's sample.  Typed keys (and GLU_HIP_SORT_DEVICE_TOP=0): the whole key's top 16 bits / the round-4 guess from the object's
 * previous attempt.
 * GLU_HIP_SORT_LDS_FINISH=0 in the environment of glu_radix_sort_create switches the attempt off.  Any pointer may be NULL. */
GLU_API glu_status glu_radix_sort_read_finish(glu_radix_sort sort, uint32_t* attempted, uint32_t* accepted,
                                              uint32_t* longest_run, uint32_t* capacity, uint32_t* top_bit);
/* Runs LONGER than the tile of the in-LDS pass (round 5; 4-byte untyped keys with values): they no longer send the whole sort back
 * to the ordinary passes -- the device lists them as segments, two segmented counting passes (the reference's pass per segment,
 * glu/RadixSort.hpp:142-182) order just their elements by the low 16 key bits, and the in-LDS pass takes every other run.  The
 * tile is the smallest enqueued one that leaves at most 8192 runs and an eighth of the pairs to those passes (failing that the
 * largest, if it leaves at most half); keys crowded into few runs are still refused.  runs / sub_blocks / pairs: what the last
 * sort gave to the segmented passes (all 0: nothing, or the sort did not end in LDS).  GLU_HIP_SORT_LONG_RUNS=0 in the environment
 * of glu_radix_sort_create restores the round-4 rule (one run longer than the largest enqueued tile refuses the sort). */
GLU_API glu_status glu_radix_sort_read_long_runs(glu_radix_sort sort, uint32_t* runs, uint32_t* sub_blocks, uint32_t* pairs);
/* The same question for the last SEGMENTED sort of the object (glu_radix_sort_run_segments_ptr; the local sort of glu_dist_*):
 * a segmented sort by 16 key bits or more first tries ONE counting pass on the top digit of those bits (into the object's scratch
 * arrays) and one pass that orders every run (segment, top digit) by the remaining bits inside LDS on its way into the output --
 * the reference's pass, glu/RadixSort.hpp:289-333, once instead of three times for 24 bits.  `tile`: pairs a workgroup orders at
 * a time; `split`: workgroups a run is split over (by ranges of the remaining key bits: runs of the sharded sort at eight ranks are
 * four tiles long); runs that come out longer than tile x split are walked in pieces.  attempted: both sequences were enqueued;
 * accepted: the device found no run longer than `capacity` (32 tiles per workgroup) and ended the sort in LDS -- otherwise the
 * ordinary segmented passes ran; runs = segments x 256.
 * GLU_HIP_SEG_LDS_FINISH=0 in the environment of glu_radix_sort_create switches the attempt off.  Waits for the device. */
GLU_API glu_status glu_radix_sort_read_seg_finish(glu_radix_sort sort, uint32_t* attempted, uint32_t* accepted,
                                                  uint32_t* longest_run, uint32_t* capacity, uint32_t* runs, uint32_t* tile,
                                                  uint32_t* split);

/* ---- exclusive scan: replaces glu::BlellochScan (glu/BlellochScan.hpp:80-191) ---------------------- */

/* BlellochScan::BlellochScan(data_type)                (BlellochScan.hpp:91-121) */
GLU_API glu_status glu_scan_create(glu_data_type data_type, glu_scan* out);
GLU_API glu_status glu_scan_destroy(glu_scan scan);
/* Optional: pre-size the internal block-sum scratch for count * num_partitions elements. */
GLU_API glu_status glu_scan_prepare(glu_scan scan, size_t count, size_t num_partitions);
/* BlellochScan::operator()(buffer, count, num_partitions)   (BlellochScan.hpp:130-139): in-place exclusive
 * `+` scan (identity 0) of num_partitions adjacent partitions of `count` elements each.  The reference's
 * argument checks are kept (:132-135): buffer != 0, count > 0, count a power of two, num_partitions >= 1. */
GLU_API glu_status glu_scan_run(glu_scan scan, glu_buffer buffer, size_t count, size_t num_partitions);
/* Raw-pointer form; additionally accepts any count > 0 (no power-of-two requirement).  From 2^23 4-byte elements up a
 * single-pass chained scan runs (not under stream capture, where the three-launch form is used): its order of additions
 * follows the arrival order of the chunks, so float sums are not bitwise reproducible from run to run there (integer
 * results are exact either way); GLU_HIP_SCAN_CHAINED=0 selects the reproducible form. */
GLU_API glu_status glu_scan_run_ptr(glu_scan scan, void* data, size_t count, size_t num_partitions, void* stream);

/* ---- reduce: replaces glu::Reduce (glu/Reduce.hpp:51-136) ----------------------------------------- */

/* Reduce::Reduce(data_type, operator)                  (Reduce.hpp:62-107) */
GLU_API glu_status glu_reduce_create(glu_data_type data_type, glu_reduce_operator op, glu_reduce* out);
GLU_API glu_status glu_reduce_destroy(glu_reduce reduce);
/* Reduce::operator()(buffer, count)                    (Reduce.hpp:111-135): data[0] = op over data[0..count),
 * component-wise for vector types.  Elements other than data[0] are left untouched (the reference clobbers
 * some of them; only data[0] is contract).  Checks kept (:113-114): buffer != 0, count > 0. */
GLU_API glu_status glu_reduce_run(glu_reduce reduce, glu_buffer buffer, size_t count);
GLU_API glu_status glu_reduce_run_ptr(glu_reduce reduce, void* data, size_t count, void* stream);

/* ---- sharded sort over the GPUs of one node ---------------------------------------------------------
 * The reference is single-device (one GL context, no communication code: SURVEY.md section 2 row C1); this is the
 * sharded form of glu::RadixSort::operator() (glu/RadixSort.hpp:273-334) that BASELINE.json configs[3] asks for.
 * One process per GPU, rank r holds slice r of the array.  A sort = stable partition of the slice by the top 8 key
 * bits -> ncclAllGather of the R x 256 bucket histograms -> identical contiguous bucket -> rank plan on every rank (a
 * bucket is never split) -> ONE grouped RCCL exchange of keys and values (receive segments in source-rank order) ->
 * local stable sort.  The ranks' outputs concatenated in rank order equal the single-device stable sort; shard sizes
 * follow the data.  If all keys of all ranks share the top byte (24-bit keys ...), the partition falls back to the next lower
 * byte, so that small-range keys still spread over the ranks; one hot bucket still bounds the balance.  RCCL is loaded with dlopen at first use (GLU_HIP_RCCL_LIB overrides the name).
 * Every rank must issue the same sequence of glu_dist calls (they contain collectives).
 * Errors are collective: when one rank finds its arguments, sizes or allocations wrong in glu_dist_sort_begin /
 * glu_dist_sort_finish / glu_dist_sort_ptr, EVERY rank returns a non-zero status from that call (the failing rank its own,
 * the others the same code with a message naming the rank) before anything is sent or received, and the object can be used
 * for the next sort; no rank is left waiting in a collective.  A rank that must give up between begin and finish (it
 * cannot allocate its receive arrays) still calls glu_dist_sort_finish, with capacity 0.  A failing RCCL or HIP call
 * after the ranks have agreed is not recoverable (it is reported by the rank that sees it; destroy the object). */

#define GLU_DIST_UNIQUE_ID_BYTES 128
/* Can this process load RCCL (GLU_HIP_RCCL_LIB, librccl.so.1, ...) with the nine entry points the sharded sort needs?
 * dlopen + dlsym only -- no device call and none of ncclGetUniqueId's side effects (a bootstrap listener socket and
 * thread per call): what a rank other than 0 asks before the ranks agree to use glu_dist_*. */
GLU_API glu_status glu_dist_available(void);
/* Rank 0: ncclGetUniqueId; the caller carries the bytes to the other ranks (MPI, torch.distributed, a file ...). */
GLU_API glu_status glu_dist_unique_id(void* id_out, size_t id_bytes);
/* ncclCommInitRank on the library's device (glu_set_device) + a local glu_radix_sort; collective over all ranks. */
GLU_API glu_status glu_dist_create(const void* unique_id, size_t id_bytes, int world_size, int rank, glu_dist* out);
GLU_API glu_status glu_dist_destroy(glu_dist dist);
GLU_API glu_status glu_dist_world(glu_dist dist, int* world_size, int* rank);
/* The bit position of the key byte the last sort on `dist` was partitioned on (24 unless the fallback ran). */
GLU_API glu_status glu_dist_partition_shift(glu_dist dist, uint32_t* shift);
/* The local glu_radix_sort that `dist` partitions and sorts with (owned by `dist`, do not destroy): for
 * glu_radix_sort_set_digit_bits / set_profiling / read_profile. */
GLU_API glu_status glu_dist_local_sorter(glu_dist dist, glu_radix_sort* out);
/* Grow-only buffers for slices of `local_count` pairs and (for glu_dist_sort_ptr) shards of `recv_capacity` pairs: after
 * it a sort whose slice / shard fit allocates nothing (the analogue of RadixSort::prepare_internal_buffers, :237-271).
 * Like glu_radix_sort_prepare it PLACES large arrays by measurement -- the local sorter's scratch, and against it the
 * send-side pair (the partitioned slice) and the receive-side pair of glu_dist_sort_ptr, three searches of 0.13-1 s each from
 * 2^27 pairs up (skipped when memory is short or GLU_HIP_SCRATCH_TUNE=0): every scatter pass of a rank's sort except the
 * first one's source, the caller's slice, then runs between pairs of arrays that were chosen, not drawn (2.33 -> 2.19 ms of
 * compute per 2^27 pairs in an alternating same-box comparison with the search switched off).  Receive arrays that a caller
 * hands to glu_dist_sort_finish are the caller's and are not placed. */
GLU_API glu_status glu_dist_prepare(glu_dist dist, size_t local_count, size_t recv_capacity);
/* First half of a sort: partition + histogram exchange + plan.  The partition is enqueued on `stream`; the call returns
 * when the host has the plan (it waits for the histogram exchange, which runs on a side stream beside the partition's
 * scatter kernel, not for the partition).  *recv_count = pairs this rank will receive. */
GLU_API glu_status glu_dist_sort_begin(glu_dist dist, const uint32_t* keys, const uint32_t* vals, size_t local_count,
                                       void* stream, size_t* recv_count);
/* Second half: the grouped exchange into the caller's arrays (capacity >= the count glu_dist_sort_begin returned) and the
 * local sort, enqueued on `stream` (no host synchronisation). */
GLU_API glu_status glu_dist_sort_finish(glu_dist dist, uint32_t* recv_keys, uint32_t* recv_vals, size_t capacity, void* stream);
/* Both halves with receive arrays owned by `dist` (grown if the shard does not fit).  LIFETIME of *out_keys / *out_vals: they
 * point into arrays `dist` owns and are valid until the NEXT glu_dist_sort_ptr call on `dist` returns -- that call may overwrite
 * them (same arrays) or outgrow them (the outgrown arrays stay allocated for exactly one more call, so a result that is still
 * being read while the next sort is enqueued does not dangle; the call after that frees them).  Copy out what must live longer;
 * glu_hip.dist.DistributedRadixSort.sort() returns tensors that alias these arrays under the same rule. */
GLU_API glu_status glu_dist_sort_ptr(glu_dist dist, const uint32_t* keys, const uint32_t* vals, size_t local_count,
                                     void* stream, uint32_t** out_keys, uint32_t** out_vals, size_t* out_count);
/* 1 if the local sort of the last sort on `dist` was the SEGMENTED sort of the low 24 bits per bucket (the shard arrives as one
 * message per source rank, each grouped by bucket; glu_radix_sort_run_segments_ptr: one segmented counting pass + an in-LDS pass
 * when the runs (bucket, next byte) fit an LDS tile -- up to four ranks x 2^27 pairs --, else three segmented passes;
 * glu_radix_sort_read_seg_finish on glu_dist_local_sorter tells which), 0 if it was the ordinary sort of all 32 bits (small
 * shards, shards made of very many tiny pieces, a partition on a lower byte). */
GLU_API glu_status glu_dist_last_local_sort(glu_dist dist, uint32_t* segmented);
/* The exchange in ROUNDS.  With more than one rank and large shards (2^24 pairs per rank and more; GLU_HIP_DIST_ROUNDS_MIN)
 * every rank's buckets are cut into `rounds` groups of about equal size (1 .. 8; default 3 from four ranks up, 1 below: at two
 * ranks half of the data never leaves and the one link to the peer bounds the exchange far above the sort; GLU_HIP_DIST_ROUNDS), round j
 * carries group j of every rank on a side stream, and the local sort of group j runs on the sort's stream behind round j
 * only: the later groups travel while the earlier ones are sorted, so ONE sort takes about partition + one round +
 * max(the other rounds, the sorts) instead of partition + exchange + sort.  The price is compute: a group is sorted by its
 * own three passes, and four sorts of a quarter cost about a third more than one sort of the whole (measured without a
 * fabric: 2.70 -> 3.2 ms per 2^27 pairs at 4 rounds).  A caller that keeps several sorts in flight on several objects, where
 * the exchange of one hides under the local sort of another anyway, sets 1 = one grouped exchange, then the local sort.
 * Every rank must use the same value (the rounds are part of the message sequence).  glu_dist_last_rounds: what the last
 * sort used. */
GLU_API glu_status glu_dist_set_rounds(glu_dist dist, int rounds);
GLU_API glu_status glu_dist_last_rounds(glu_dist dist, uint32_t* rounds);
/* CUs that the sort kernels of `dist` leave free (for RCCL kernels of another sort in flight; the partition pass always
 * leaves 8, one per XCD, for the histogram all-gather that runs beside its scatter kernel). */
GLU_API glu_status glu_dist_set_reserved_cus(glu_dist dist, int cus);
/* Device time per phase, averaged over the sorts since the last call (after glu_dist_set_profiling(dist, 1)):
 * ms4 = {partition, histogram exchange + plan (side stream), exchange, local sort}. */
GLU_API glu_status glu_dist_set_profiling(glu_dist dist, int enable);
GLU_API glu_status glu_dist_phase_times(glu_dist dist, double* ms4, uint64_t* sorts);
/* The plan as pure host functions (no device, no RCCL needed: unit-testable with simulated ranks).
 * all_hist: [world_size][256] bucket histograms; bucket_owner: [256] rank of every bucket (contiguous, monotone, balanced
 * on the boundary nearest to r * N / R); send_counts / recv_counts: [world_size] for `rank`. */
GLU_API glu_status glu_dist_plan_buckets(const uint32_t* all_hist, int world_size, int* bucket_owner);
GLU_API glu_status glu_dist_plan_counts(const uint32_t* all_hist, int world_size, int rank, const int* bucket_owner,
                                        uint64_t* send_counts, uint64_t* recv_counts);
/* The groups of the exchange in rounds (glu_dist_set_rounds): group_cut: [world_size][rounds + 1], group j of rank q =
 * the buckets [group_cut[q][j], group_cut[q][j + 1]) -- contiguous, in order, together exactly q's buckets, cut on the bucket
 * boundaries nearest to j / rounds of what q receives (a bucket is never split; groups may be empty). */
GLU_API glu_status glu_dist_plan_groups(const uint32_t* all_hist, int world_size, const int* bucket_owner, int rounds,
                                        int* group_cut);

/* ---- timing: replaces glu::measure_gl_elapsed_time (glu/gl_utils.hpp:249-265) ---------------------- */

/* glGenQueries + glBeginQuery(GL_TIME_ELAPSED) on the library queue. */
GLU_API glu_status glu_timer_begin(glu_timer* out);
/* glEndQuery + glGetQueryObjectui64v(GL_QUERY_RESULT): waits, returns nanoseconds, frees the timer. */
GLU_API glu_status glu_timer_end(glu_timer timer, uint64_t* elapsed_ns);

#ifdef __cplusplus
}
#endif

#endif /* GLU_HIP_H */
