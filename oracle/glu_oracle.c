/*
 * glu_oracle.c -- CPU restatement of the loryruta/gl-radix-sort (v2) algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The shipped path (gl-radix-sort_amd/) never links,
 * imports or calls it and fails loudly without its HIP library.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   - radix count pass        glu/RadixSort.hpp:33-57    (k_radix_sort_counting_shader main)
 *   - radix reorder pass      glu/RadixSort.hpp:102-182  (prefix_sum + main of the reorder shader)
 *   - sort driver             glu/RadixSort.hpp:273-334  (operator(): early-out, ping-pong, num_steps)
 *   - scratch sizing          glu/RadixSort.hpp:337-353
 *   - scan up/down sweep      glu/BlellochScan.hpp:26-45, 59-75 (shaders), :142-190 (host loops)
 *   - reduce                  glu/Reduce.hpp:24-37 (shader), :111-135 (host loop)
 *   - integer helpers         glu/gl_utils.hpp:279-302
 *   - test input generator    test/util/Random.hpp:15-39 (std::minstd_rand)
 *
 * PARITY PINNING.  The reference's device code is GLSL compiled at run time by an OpenGL 4.6 driver;
 * it cannot be built or run in this environment (no GL context, no GLSL compiler, Catch2/glfw/glm
 * submodules are empty), so there is no oracle/_ref build.  This restatement is pinned against every
 * known-answer vector the reference's own tests hold (tests/golden/reference_vectors.json:
 * Reduce-simple-uint, Reduce-all, BlellochScan-simple, minstd_rand conformance) and against the
 * reference tests' own assertions (sorted + permutation on its seeded inputs; scan == exclusive scan).
 * The reference has NO test that checks the order of values of equal keys, so *stability* is pinned only
 * by this literal restatement of the shader index arithmetic ("parity unpinned" for value order by any
 * reference-run output).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GLU_ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * gl_utils.hpp:279-302 integer helpers
 * ---------------------------------------------------------------------------------------------- */

/* gl_utils.hpp:279-283 (the reference detours through double; exact for the sizes used here) */
GLU_ORACLE_API uint64_t glu_oracle_div_ceil(uint64_t n, uint64_t d)
{
    double q = (double) n / (double) d;
    uint64_t f = (uint64_t) q;
    return ((double) f < q) ? f + 1 : f;
}

/* gl_utils.hpp:285-289 -- note: true for 0 */
GLU_ORACLE_API int glu_oracle_is_power_of_2(uint64_t n) { return (n & (n - 1)) == 0; }

/* gl_utils.hpp:291-302 -- shifts only up to 16, i.e. valid for values <= 2^32 */
GLU_ORACLE_API uint64_t glu_oracle_next_power_of_2(uint64_t n)
{
    n--;
    n |= n >> 1;
    n |= n >> 2;
    n |= n >> 4;
    n |= n >> 8;
    n |= n >> 16;
    n++;
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * test/util/Random.hpp:15-39 -- std::minstd_rand (a = 48271, m = 2^31 - 1), seed 0 -> default seed 1
 * sample_int(min, max) = engine() % (max - min) + min
 * ---------------------------------------------------------------------------------------------- */
GLU_ORACLE_API void glu_oracle_minstd_sample(uint64_t seed, uint64_t n, uint32_t min, uint32_t max, uint32_t* out)
{
    uint64_t state = seed % 2147483647ull;
    if (state == 0) state = 1; /* std::linear_congruential_engine: seed == 0 (mod m) with c == 0 -> 1 */
    uint32_t range = max - min;
    for (uint64_t i = 0; i < n; i++)
    {
        state = (state * 48271ull) % 2147483647ull;
        out[i] = (uint32_t) ((uint32_t) state % range) + min;
    }
}

/* ------------------------------------------------------------------------------------------------
 * BlellochScan (uint32, OPERATION = +, IDENTITY = 0), literal per-dispatch restatement.
 * ---------------------------------------------------------------------------------------------- */

/* One upsweep dispatch, BlellochScan.hpp:26-45.  Invocations exist for thread_i in
 * [0, num_workgroups * 1024); lval = subgroupShuffleUp(data[i], 1) is the previous invocation's data[i],
 * i.e. data[i - u_step]; only odd lanes write, the last element of the partition is set to IDENTITY. */
static void scan_upsweep_dispatch(uint32_t* data, uint32_t count, uint32_t step, uint64_t num_threads,
                                  uint32_t num_partitions)
{
    for (uint32_t p = 0; p < num_partitions; p++)
    {
        uint32_t end_i = (p + 1) * count;
        for (uint64_t t = 0; t < num_threads; t++)
        {
            uint32_t i = p * count + (uint32_t) t * step + step - 1;
            if (i < end_i)
            {
                if (i == end_i - 1)
                    data[i] = 0;
                else if (t % 2 == 1)
                    data[i] = data[i] + data[i - step];
            }
        }
    }
}

/* One downsweep dispatch, BlellochScan.hpp:59-75 (all arithmetic in uint32, wrap-around included:
 * with u_step == 0 the index i underflows exactly as in the shader). */
static void scan_downsweep_dispatch(uint32_t* data, uint32_t count, uint32_t step, uint64_t num_threads,
                                    uint32_t num_partitions, uint64_t buffer_len)
{
    for (uint32_t p = 0; p < num_partitions; p++)
    {
        uint32_t end_i = (p + 1) * count;
        for (uint64_t x = 0; x < num_threads; x++)
        {
            uint32_t i = p * count + (uint32_t) x * (step << 1) + (step - 1);
            uint32_t next_i = i + step;
            if (next_i < end_i)
            {
                if (i >= buffer_len || next_i >= buffer_len) continue; /* never hit for valid inputs */
                uint32_t tmp = data[i];
                data[i] = data[next_i];
                data[next_i] = data[next_i] + tmp;
            }
            else if (i < end_i)
            {
                data[i] = 0;
            }
        }
    }
}

/* BlellochScan::operator(), BlellochScan.hpp:130-190.  Returns 0, or -1 on the reference's argument
 * checks (:132-135) failing (where the reference would print and exit(1)). */
GLU_ORACLE_API int glu_oracle_blelloch_scan_u32(uint32_t* data, uint64_t count, uint64_t num_partitions)
{
    if (!data || count == 0 || !glu_oracle_is_power_of_2(count) || num_partitions < 1) return -1;

    /* upsweep, :149-165 */
    int step = 1;
    int level_count = (int) count;
    for (;;)
    {
        uint64_t num_workgroups = glu_oracle_div_ceil((uint64_t) level_count, 1024);
        scan_upsweep_dispatch(data, (uint32_t) count, (uint32_t) step, num_workgroups * 1024, (uint32_t) num_partitions);
        step <<= 1;
        level_count >>= 1;
        if (level_count <= 1) break;
    }

    /* downsweep, :175-189 */
    step = (int) (glu_oracle_next_power_of_2((uint64_t) (int) count) >> 1);
    uint64_t lc = 1;
    for (;;)
    {
        uint64_t num_workgroups = glu_oracle_div_ceil(lc, 1024);
        scan_downsweep_dispatch(data, (uint32_t) count, (uint32_t) step, num_workgroups * 1024,
                                (uint32_t) num_partitions, count * num_partitions);
        step >>= 1;
        lc <<= 1;
        if (step == 0) break;
    }
    return 0;
}

/* std::exclusive_scan equivalent the reference's tests compare against (blelloch_scan_tests.cpp:43-45). */
GLU_ORACLE_API void glu_oracle_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t count,
                                                  uint64_t num_partitions)
{
    for (uint64_t p = 0; p < num_partitions; p++)
    {
        uint32_t acc = 0;
        for (uint64_t i = 0; i < count; i++)
        {
            uint32_t v = in[p * count + i];
            out[p * count + i] = acc;
            acc += v;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * RadixSort, literal restatement
 * ---------------------------------------------------------------------------------------------- */

#define ORACLE_NUM_THREADS 1024u

/* Count dispatch, RadixSort.hpp:33-57.  block_count is radix-major: [radix * nbp2 + block]. */
static void radix_count_dispatch(const uint32_t* keys, uint32_t* block_count, uint32_t* global_count, uint32_t count,
                                 uint32_t shift, uint32_t nbp2, uint32_t num_blocks)
{
    for (uint32_t wg = 0; wg < num_blocks; wg++)
    {
        for (uint32_t radix = 0; radix < 16; radix++) block_count[radix * nbp2 + wg] = 0; /* :35-38 */
        for (uint32_t t = 0; t < ORACLE_NUM_THREADS; t++)
        {
            uint32_t i = wg * ORACLE_NUM_THREADS + t;
            if (i < count)
            {
                uint32_t radix = (keys[i] >> shift) & 0xf; /* :46 */
                block_count[radix * nbp2 + wg] += 1;        /* :47 */
            }
        }
        for (uint32_t l = 0; l < 16; l++) global_count[l] += block_count[l * nbp2 + wg]; /* :52-56 */
    }
}

/* prefix_sum() of the reorder shader, RadixSort.hpp:102-140: 1024-wide Blelloch scan in shared memory. */
static void radix_block_prefix_sum(uint32_t* s)
{
    for (uint32_t step = 1; step < ORACLE_NUM_THREADS; step <<= 1) /* upsweep :107-119 */
    {
        for (uint32_t t = 1; t < ORACLE_NUM_THREADS; t += 2)
        {
            uint64_t i = (uint64_t) t * step + (step - 1);
            if (i < ORACLE_NUM_THREADS) s[i] = s[i] + s[i - step];
        }
    }
    s[ORACLE_NUM_THREADS - 1] = 0; /* :122 */
    for (uint32_t step = ORACLE_NUM_THREADS >> 1; step > 0; step >>= 1) /* downsweep :127-139 */
    {
        for (uint32_t t = 0; t < ORACLE_NUM_THREADS; t += 2)
        {
            uint64_t i = (uint64_t) t * step + (step - 1);
            if (i + step < ORACLE_NUM_THREADS)
            {
                uint32_t tmp = s[i];
                s[i] = s[i + step];
                s[i + step] = tmp + s[i + step];
            }
        }
    }
}

/* Reorder dispatch, RadixSort.hpp:142-182. */
static void radix_reorder_dispatch(const uint32_t* src_key, const uint32_t* src_val, uint32_t* dst_key,
                                   uint32_t* dst_val, const uint32_t* block_offset, const uint32_t* global_count,
                                   uint32_t count, uint32_t shift, uint32_t nbp2, uint32_t num_blocks)
{
    uint32_t global_offset[16]; /* subgroupExclusiveAdd over the 16 global counts, :148-152 */
    uint32_t acc = 0;
    for (uint32_t d = 0; d < 16; d++)
    {
        global_offset[d] = acc;
        acc += global_count[d];
    }

    uint32_t s[ORACLE_NUM_THREADS];
    for (uint32_t wg = 0; wg < num_blocks; wg++)
    {
        for (uint32_t radix = 0; radix < 16; radix++) /* :157-181 */
        {
            for (uint32_t t = 0; t < ORACLE_NUM_THREADS; t++)
            {
                uint32_t i = wg * ORACLE_NUM_THREADS + t;
                int should_place = 0;
                if (i < count) should_place = ((src_key[i] >> shift) & 0xf) == radix;
                s[t] = should_place ? 1 : 0;
            }
            radix_block_prefix_sum(s);
            for (uint32_t t = 0; t < ORACLE_NUM_THREADS; t++)
            {
                uint32_t i = wg * ORACLE_NUM_THREADS + t;
                if (i < count && ((src_key[i] >> shift) & 0xf) == radix)
                {
                    uint32_t di = global_offset[radix] + block_offset[radix * nbp2 + wg] + s[t]; /* :174-177 */
                    dst_key[di] = src_key[i];
                    dst_val[di] = src_val[i];
                }
            }
        }
    }
}

/* required_*_size, RadixSort.hpp:337-353 (bytes). */
GLU_ORACLE_API uint64_t glu_oracle_radix_block_count_buffer_size(uint64_t count)
{
    uint64_t num_blocks = glu_oracle_div_ceil(count, 1024);
    uint64_t nbp2 = glu_oracle_next_power_of_2(num_blocks);
    return glu_oracle_next_power_of_2(16 * nbp2) * 4;
}
GLU_ORACLE_API uint64_t glu_oracle_radix_scratch_buffer_size(uint64_t count)
{
    return glu_oracle_next_power_of_2(count) * 4;
}

/*
 * RadixSort::operator(), RadixSort.hpp:273-334, restated literally.
 *   key, val           the caller's buffers (sorted "in place" exactly as far as the reference does)
 *   key_scratch, val_scratch   the instance's internal scratch buffers (count entries each, caller-allocated
 *                      here so that a test can inspect them)
 *   trace_block_count  optional [num_passes][16 * nbp2]: the *scanned* table of every pass
 *   result_in_scratch  out: 1 when the reference leaves the final pass output in the scratch buffers (odd
 *                      number of executed passes), 0 when it is in key/val
 * Returns the number of passes executed, or -1 for an argument the reference rejects (:275-276).
 */
GLU_ORACLE_API int glu_oracle_radix_sort_reference(uint32_t* key, uint32_t* val, uint32_t* key_scratch,
                                                   uint32_t* val_scratch, uint64_t count, uint64_t num_steps,
                                                   uint32_t* trace_block_count, int* result_in_scratch)
{
    if (result_in_scratch) *result_in_scratch = 0;
    if (!key || !val) return -1;
    if (count <= 1) return 0; /* :278-279 */

    uint32_t num_blocks = (uint32_t) glu_oracle_div_ceil(count, 1024);
    uint32_t nbp2 = (uint32_t) glu_oracle_next_power_of_2(num_blocks);
    uint64_t table_len = glu_oracle_radix_block_count_buffer_size(count) / 4;

    uint32_t* block_count = (uint32_t*) malloc(table_len * sizeof(uint32_t));
    uint32_t global_count[16];
    uint32_t* key_buffers[2] = {key, key_scratch};
    uint32_t* val_buffers[2] = {val, val_scratch};

    int step = 0;
    for (; step < 8;)
    {
        memset(block_count, 0, table_len * sizeof(uint32_t)); /* :293 */
        memset(global_count, 0, sizeof(global_count));        /* :294 */

        radix_count_dispatch(key_buffers[step % 2], block_count, global_count, (uint32_t) count,
                             (uint32_t) step << 2, nbp2, num_blocks); /* :296-307 */

        glu_oracle_blelloch_scan_u32(block_count, nbp2, 16); /* :311 */
        if (trace_block_count) memcpy(trace_block_count + (size_t) step * 16 * nbp2, block_count, (size_t) 16 * nbp2 * 4);

        radix_reorder_dispatch(key_buffers[step % 2], val_buffers[step % 2], key_buffers[(step + 1) % 2],
                               val_buffers[(step + 1) % 2], block_count, global_count, (uint32_t) count,
                               (uint32_t) step << 2, nbp2, num_blocks); /* :315-329 */

        ++step;
        if ((uint64_t) step == num_steps || step == 8) break; /* :331-332 */
    }
    free(block_count);
    if (result_in_scratch) *result_in_scratch = step % 2;
    return step;
}

/* ------------------------------------------------------------------------------------------------
 * Independent checker: stable sort by the low `key_bits` bits of the key (what the reference's algorithm
 * amounts to: SURVEY.md section 8c).  LSD byte-wise counting sort, O(n), used for sizes where the literal
 * restatement above is too slow, and as the CPU baseline "port" in bench.py.
 * ---------------------------------------------------------------------------------------------- */
GLU_ORACLE_API int glu_oracle_stable_sort_pairs_u32(uint32_t* key, uint32_t* val, uint64_t count, uint32_t key_bits)
{
    if (count <= 1 || key_bits == 0) return 0;
    if (key_bits > 32) key_bits = 32;
    uint32_t* k2 = (uint32_t*) malloc(count * 4);
    uint32_t* v2 = (uint32_t*) malloc(count * 4);
    if (!k2 || !v2) { free(k2); free(v2); return -1; }
    uint32_t *ks = key, *vs = val, *kd = k2, *vd = v2;
    for (uint32_t shift = 0; shift < key_bits; shift += 8)
    {
        uint32_t bits = key_bits - shift < 8 ? key_bits - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        uint64_t hist[256];
        memset(hist, 0, sizeof(hist));
        for (uint64_t i = 0; i < count; i++) hist[(ks[i] >> shift) & mask]++;
        uint64_t acc = 0;
        for (uint32_t d = 0; d < 256; d++) { uint64_t c = hist[d]; hist[d] = acc; acc += c; }
        for (uint64_t i = 0; i < count; i++)
        {
            uint64_t p = hist[(ks[i] >> shift) & mask]++;
            kd[p] = ks[i];
            vd[p] = vs[i];
        }
        uint32_t* t;
        t = ks; ks = kd; kd = t;
        t = vs; vs = vd; vd = t;
    }
    if (ks != key)
    {
        memcpy(key, ks, count * 4);
        memcpy(val, vs, count * 4);
    }
    free(k2);
    free(v2);
    return 0;
}

/* Same for 64-bit keys + 32-bit payload (BASELINE.json config 5; the reference has no 64-bit path, the
 * contract is the same stable order). */
GLU_ORACLE_API int glu_oracle_stable_sort_pairs_u64(uint64_t* key, uint32_t* val, uint64_t count, uint32_t key_bits)
{
    if (count <= 1 || key_bits == 0) return 0;
    if (key_bits > 64) key_bits = 64;
    uint64_t* k2 = (uint64_t*) malloc(count * 8);
    uint32_t* v2 = (uint32_t*) malloc(count * 4);
    if (!k2 || !v2) { free(k2); free(v2); return -1; }
    uint64_t *ks = key, *kd = k2;
    uint32_t *vs = val, *vd = v2;
    for (uint32_t shift = 0; shift < key_bits; shift += 8)
    {
        uint32_t bits = key_bits - shift < 8 ? key_bits - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        uint64_t hist[256];
        memset(hist, 0, sizeof(hist));
        for (uint64_t i = 0; i < count; i++) hist[(ks[i] >> shift) & mask]++;
        uint64_t acc = 0;
        for (uint32_t d = 0; d < 256; d++) { uint64_t c = hist[d]; hist[d] = acc; acc += c; }
        for (uint64_t i = 0; i < count; i++)
        {
            uint64_t p = hist[(ks[i] >> shift) & mask]++;
            kd[p] = ks[i];
            vd[p] = vs[i];
        }
        uint64_t* t = ks; ks = kd; kd = t;
        uint32_t* u = vs; vs = vd; vd = u;
    }
    if (ks != key)
    {
        memcpy(key, ks, count * 8);
        memcpy(val, vs, count * 4);
    }
    free(k2);
    free(v2);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Reduce, literal restatement for uint32 (Reduce.hpp:24-37 shader, :111-135 host loop), with the subgroup
 * size as a parameter: the shader hard-wires a stride of 32 per depth (`1 << (5 * u_depth)`), so it is only
 * correct for subgroup_size == 32; that is the configuration the reference's published numbers and tests ran on.
 * op: 0 sum, 1 mul, 2 min, 3 max (ReduceOperator, Reduce.hpp:42-48).  Result is data[0]; the rest of the
 * buffer is clobbered exactly as the shader clobbers it.
 * ---------------------------------------------------------------------------------------------- */
static uint32_t reduce_op_u32(uint32_t a, uint32_t b, int op)
{
    switch (op)
    {
    case 0: return a + b;
    case 1: return a * b;
    case 2: return a < b ? a : b;
    default: return a > b ? a : b;
    }
}

GLU_ORACLE_API int glu_oracle_reduce_reference_u32(uint32_t* data, uint64_t count, int op, uint32_t subgroup_size)
{
    if (!data || count == 0) return -1; /* :113-114 */
    for (int depth = 0;; depth++)
    {
        int step = 1 << (5 * depth);
        if ((uint64_t) step >= count) break; /* :123-124 */
        uint64_t level_count = count >> (5 * depth);
        uint64_t num_threads = glu_oracle_div_ceil(level_count, 1024) * 1024;
        for (uint64_t sg = 0; sg < num_threads; sg += subgroup_size)
        {
            int have = 0;
            uint32_t r = 0;
            for (uint32_t l = 0; l < subgroup_size; l++) /* subgroup op over the active lanes, :27-31 */
            {
                uint64_t i = (sg + l) * (uint64_t) step;
                if (i < count)
                {
                    r = have ? reduce_op_u32(r, data[i], op) : data[i];
                    have = 1;
                }
            }
            uint64_t i0 = sg * (uint64_t) step;
            if (have && i0 < count) data[i0] = r; /* lane 0 writes back, :32-35 */
        }
    }
    return 0;
}
