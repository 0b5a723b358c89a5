// radix_pair_passes.hpp -- one read of the keys for the digit tables of TWO counting passes (large sorts).
//
// The count pass (k_radix_sort_counting_shader, reference glu/RadixSort.hpp:11-58) reads every key once per pass: 4 x 0.165
// ms of a 4.3 ms sort of 2^28 pairs.  The table of pass p + 1 cannot be counted before pass p has run, because its blocks
// are ranges of pass p's OUTPUT -- unless the blocks of pass p + 1 are chosen to be what pass p's output is made of:
//
//   pass p ("leader") cuts its input into nb blocks; its output is the concatenation, digit value d major and block b
//   minor, of the UNITS (d, b) = the keys of block b whose digit p is d, in input order.  A unit's position and length are
//   the scanned count table of pass p.  A pass p + 1 ("follower") whose workgroup w takes a run of whole units
//   [U_w, U_w+1) needs, per digit value e of digit p + 1, the number of keys with that value in its run = the sum over its
//   units of T2[d][b][e] = #keys of block b with digit p = d and digit p + 1 = e: a two-digit histogram per block, which
//   the leader's count kernel builds from the keys it is reading anyway (256 x 256 16-bit counters = 128 KiB of LDS).
//
// radix_pair_count_kernel: the leader's count kernel, T1 (the usual table) and T2.
// radix_pair_unitsum_kernel: instead of the follower's count kernel: unit runs balanced on elements, their digit counts
// summed from T2 (64 MiB of table traffic per pair of passes instead of 1 GiB of keys).
// The follower's row scan and scatter are the usual kernels; the scatter takes its element range from `ranges`.
//
// Three things send a follower back to its own count kernel (PassPlan::pair_fallback, decided on the device; the launch
// sequence is the same either way, the kernels that are not needed return at once):
//   * a 16-bit counter of T2 overflowed (more than 65535 keys of one block share both digit values): found by comparing
//     every row sum of T2 with T1;
//   * a unit is longer than 1/16 of a workgroup's share: runs of whole units could not be balanced (a digit value that
//     holds more than 6 % of a block's keys);
//   * a workgroup's run has more than 2048 units (a region of very rare digit values): summing their rows in one
//     workgroup would take longer than counting.
#pragma once

#include "radix_sort_kernels.hpp"

namespace glu_hip
{
constexpr uint32_t kPairRadix = 256;               // 8-bit digits only
constexpr uint32_t kPairRowWords = kPairRadix / 2; // a T2 row: 256 16-bit counters in 128 words
constexpr uint32_t kPairMaxRunUnits = 2048;       // a follower workgroup sums at most this many T2 rows (1 MiB)

// physical LDS word of the counter pair holding (d, e): the row's words are permuted by d so that a wave whose keys share
// one of the two digit values still spreads over all banks
__device__ __forceinline__ uint32_t pair_word(uint32_t d, uint32_t e) { return d * kPairRowWords + ((e >> 1) ^ (d & 63u)); }

struct PairCountSmem
{
    static constexpr int WAVES = 16;
    uint32_t hist1[WAVES][kPairRadix];        // wave-private counters of digit p (as in radix_count_kernel)
    uint32_t hist2[kPairRadix * kPairRowWords]; // shared 16-bit counters of (digit p, digit p + 1)
};

// WIDE ROWS (round 6).  A 16-bit counter of T2 wraps when more than 65535 keys of one block share both digit values -- a key value
// that holds more than 6 % of the input: three distinct values, 10 % zeros.  Round 5 could only notice (the row's counters do not add
// up to T1) and the sort that tried to end in LDS was refused for want of exact run lengths.  With `wide` (the leader of such an
// attempt) the block now puts its rows right: the rows that do not add up (at most kPairWideRows per block: a block of 2^20 keys has
// at most 15 counters beyond 65535) are counted AGAIN with 32-bit counters -- a second read of the block's keys, 4 MiB from L2 /
// HBM, only in blocks that have such a row --, the T2 row is rewritten with the low 16 bits of the exact counts and the high 16
// bits go to the block's wide rows:
//   wide[b * kPairWideStride]                  how many wide rows block b has (~0: more than kPairWideRows, left as they were)
//   wide[b * kPairWideStride + 1 + k]          the first digit d of its k-th wide row
//   wide[b * kPairWideStride + 16 + k * 256 + e]   count >> 16 of (d, e)
// radix_finish_lengths_kernel adds 65536 x those.  The follower of the pair still counts for itself (its unit sums need whole rows).
constexpr uint32_t kPairWideRows = 14;
constexpr uint32_t kPairWideStride = 16 + kPairWideRows * 256;

// t2 row of unit (d, b): words [(d * nb + b) * 128, + 128), counter e in the low (e even) / high (e odd) half of word e / 2
template<typename KeyT, int TILE, bool XF = false, bool COLLECT = false>
__global__ __launch_bounds__(1024) void radix_pair_count_kernel(const KeyT* __restrict__ keys_a, uint32_t* __restrict__ table,
                                                                uint32_t* __restrict__ t2, uint32_t n, uint32_t shift,
                                                                uint32_t mask, uint32_t shift2, uint32_t mask2,
                                                                uint32_t tiles_total, uint32_t xform, const KeyT* keys_b,
                                                                PassPlan* plan, uint32_t pass, uint32_t plan_flags = 0,
                                                                uint32_t share = 0, uint32_t* __restrict__ wide = nullptr)
{
    constexpr int THREADS = 1024;
    constexpr int WAVES = PairCountSmem::WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PairCountSmem& s = *reinterpret_cast<PairCountSmem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KeyT* __restrict__ keys = keys_a;
    if (plan)
    {
        shift -= plan->shift_down[pass]; // (PassPlan::top_bit)
        shift2 -= plan->shift_down[pass + 1];
        // this pass's digit lies in key bits that do not vary: an identity, known without counting -- no tables either, so
        // the follower counts for itself (or finds its own digit constant)
        if (plan->off[pass] || ((plan_flags & kPlanShortcut) && plan_digit_is_constant(plan, shift, mask))) // (kernel-uniform; off: radix_lds_finish.hpp)
        {
            if (blockIdx.x == 0 && tid == 0)
            {
                plan->skip[pass] = kSkipWithoutCounting;
                plan->pair_fallback[pass + 1] = 1;
            }
            return;
        }
        if (pass > 0 && plan->flip[pass]) keys = keys_b;
        if (blockIdx.x == 0 && tid == 0)
        {
            plan->skip[pass] = 0;
            if (COLLECT && (plan_flags & kPlanCollectBits)) plan->bits_valid = 1;
        }
    }
    KeyT acc_or = 0, acc_and = (KeyT) ~(KeyT) 0; // of the raw keys this thread reads
    for (uint32_t i = tid; i < sizeof(PairCountSmem) / 4; i += THREADS) reinterpret_cast<uint32_t*>(&s)[i] = 0;
    __syncthreads();

    uint64_t begin, end;
    block_range(blockIdx.x, gridDim.x, tiles_total, TILE, n, share, begin, end);
    uint32_t* my_hist = s.hist1[wave];
    const KeyCodec<KeyT, XF> codec_in(xform & 3u);

    // every lane of the wave is active when this runs
    auto both_digits = [&](KeyT raw) {
        const KeyT k = codec_in.encode(raw);
        return digit_of<KeyT>(k, shift, mask) | (digit_of<KeyT>(k, shift2, mask2) << 8);
    };
    constexpr int VEC = 16 / sizeof(KeyT);
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint64_t nvec = vec_ok ? (end - begin) / VEC : 0;
    const VecT* vkeys = reinterpret_cast<const VecT*>(keys + begin);
    // The block's keys, every pair of digit values handed to add(d | e << 8, how many): groups of equal values with one call when few
    // values dominate a wave (wave_tally).  Called once for the tables, and once more for the wide rows of a block that needs them.
    auto count_keys = [&](auto&& add, auto collect) {
        constexpr bool kCollect = decltype(collect)::value;
        TallyRun run;
        auto tally = [&](auto peel, KeyT raw) { wave_tally_mode(peel, both_digits(raw), lane, run, add); };
        auto tally_one = [&](KeyT raw) { // lanes may be inactive
            if (kCollect) acc_or |= raw, acc_and &= raw;
            add(both_digits(raw), 1u);
        };
        auto tally_vec = [&](auto peel, const VecT& a) {
            if constexpr (sizeof(KeyT) == 4)
            {
                if (kCollect) acc_or |= a.x | a.y | a.z | a.w, acc_and &= a.x & a.y & a.z & a.w;
                tally(peel, a.x); tally(peel, a.y); tally(peel, a.z); tally(peel, a.w);
            }
            else
            {
                if (kCollect) acc_or |= a.x | a.y, acc_and &= a.x & a.y;
                tally(peel, a.x); tally(peel, a.y);
            }
        };
        uint64_t vbase = 0;
        // (the loop twice, chosen once per wave from its first keys: see wave_tally.  An explicit prefetch of the next
        // iteration's four vectors made the kernel 8 % slower)
        auto main_loop = [&](auto peel) {
            for (; vbase + 4 * THREADS <= nvec; vbase += 4 * THREADS) // block-uniform trip count, 4 x 16 B in flight per lane
            {
                if (wave_tally_gives_up(peel, run)) break; // (peel mode only; the wave comes back in the stateless mode)
                VecT a = load_streaming(&vkeys[vbase + tid]);
                VecT b = load_streaming(&vkeys[vbase + tid + THREADS]);
                VecT c = load_streaming(&vkeys[vbase + tid + 2 * THREADS]);
                VecT d = load_streaming(&vkeys[vbase + tid + 3 * THREADS]);
                tally_vec(peel, a);
                tally_vec(peel, b);
                tally_vec(peel, c);
                tally_vec(peel, d);
            }
        };
        if (4 * THREADS <= nvec) wave_tally_dispatch(both_digits(vkeys[tid].x), lane, run, main_loop, add);
        uint64_t i = begin + vbase * VEC + tid;
        for (; i + 7ull * THREADS < end; i += 8ull * THREADS)
        {
            KeyT k[8];
#pragma unroll
            for (int j = 0; j < 8; j++) k[j] = keys[i + (uint64_t) j * THREADS];
#pragma unroll
            for (int j = 0; j < 8; j++) tally_one(k[j]);
        }
        for (; i < end; i += THREADS) tally_one(keys[i]);
    };
    auto add_count = [&](uint32_t de, uint32_t c) {
        atomicAdd(&my_hist[de & 255u], c);
        atomicAdd(&s.hist2[pair_word(de & 255u, de >> 8)], c << (16u * ((de >> 8) & 1u)));
    };
    count_keys(add_count, std::integral_constant<bool, COLLECT>());
    __syncthreads();
    if (COLLECT && (plan_flags & kPlanCollectBits)) plan_publish_bits<KeyT>(plan, acc_or, acc_and, lane);

    // T1: the usual table entry; kept in hist1[0] for the row check below
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    if (tid < kPairRadix)
    {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) c += s.hist1[w][tid];
        table[(size_t) tid * nb + b] = c;
        s.hist1[0][tid] = c; // (thread tid is the only reader of column tid)
    }
    __syncthreads();
    // T2: every wave writes the rows d = wave, wave + 16, ...; a row whose counters do not add up to T1[d] had an overflow
    bool bad = false;
    for (uint32_t d = wave; d < kPairRadix; d += WAVES)
    {
        const uint32_t j0 = lane ^ (d & 63u); // the counter pair this lane's two physical words hold: j0 and 64 + j0
        const uint32_t w0 = s.hist2[d * kPairRowWords + lane], w1 = s.hist2[d * kPairRowWords + 64 + lane];
        uint32_t sum = (w0 & 0xFFFFu) + (w0 >> 16) + (w1 & 0xFFFFu) + (w1 >> 16);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        bad = bad || sum != s.hist1[0][d];
        uint32_t* row = t2 + ((size_t) d * nb + b) * kPairRowWords;
        row[j0] = w0;
        row[64 + j0] = w1;
    }
    if (bad && lane == 0 && plan) plan->pair_fallback[pass + 1] = 1;
    if (!wide) return; // (kernel-uniform)
    // ---- wide rows: which rows did not add up (re-checked from T1 and the row in LDS, now that every T1 entry is in hist1[0])
    __shared__ uint32_t nwide, wide_d[kPairWideRows];
    uint32_t* const slot_of = s.hist1[1];              // [256]: 1 + the wide row of digit value d, 0: none
    uint32_t* const wide_cnt = &s.hist1[2][0];         // [kPairWideRows][256] exact 32-bit counts (hist1 holds 16 rows of 256 words)
    static_assert(2 + kPairWideRows <= PairCountSmem::WAVES, "the wide counters live in the wave-private T1 rows, which are done with");
    __syncthreads(); // (every wave has read its T1 rows and checked its T2 rows)
    for (uint32_t i = tid; i < (1u + kPairWideRows) * kPairRadix; i += THREADS) s.hist1[1][i] = 0u;
    if (tid == 0) nwide = 0;
    __syncthreads();
    for (uint32_t d = wave; d < kPairRadix; d += WAVES)
    {
        const uint32_t w0 = s.hist2[d * kPairRowWords + lane], w1 = s.hist2[d * kPairRowWords + 64 + lane];
        uint32_t sum = (w0 & 0xFFFFu) + (w0 >> 16) + (w1 & 0xFFFFu) + (w1 >> 16);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (sum != s.hist1[0][d] && lane == 0)
        {
            const uint32_t k = atomicAdd(&nwide, 1u);
            if (k < kPairWideRows) slot_of[d] = k + 1u, wide_d[k] = d;
        }
    }
    __syncthreads();
    const uint32_t nw = nwide; // (workgroup-uniform)
    uint32_t* const hdr = wide + (size_t) b * kPairWideStride;
    // (a block whose keys are all ONE value -- all-zero keys, the reference's benchmark input: its row cannot be put right for less
    // than a second read, and a sort of equal keys is better off refused: the ordinary passes then skip on the bits the first read
    // collected, 0.31 ms for 2^28 keys.  Known where the key bits are collected: untyped keys.  The waves' OR / AND meet in LDS.)
    bool one_value = false;
    if (COLLECT && nw != 0)
    {
        KeyT o = acc_or, a = acc_and;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1)
        {
            o |= (KeyT) __shfl_xor((unsigned long long) o, sh);
            a &= (KeyT) __shfl_xor((unsigned long long) a, sh);
        }
        KeyT* const wave_bits = reinterpret_cast<KeyT*>(&s.hist2[0]); // (the T2 rows have left for global memory)
        __syncthreads();
        if (lane == 0) wave_bits[2 * wave] = o, wave_bits[2 * wave + 1] = a;
        __syncthreads();
        o = 0, a = (KeyT) ~(KeyT) 0;
        for (int w = 0; w < WAVES; w++) o |= wave_bits[2 * w], a &= wave_bits[2 * w + 1];
        one_value = o == a;
        __syncthreads();
    }
    if (nw == 0 || nw > kPairWideRows || one_value)
    {
        if (tid == 0) hdr[0] = nw == 0 ? 0u : 0xFFFFFFFFu;
        return;
    }
    // the block's keys again, exact counts of the rows that wrapped
    auto add_wide = [&](uint32_t de, uint32_t c) {
        const uint32_t sl = slot_of[de & 255u];
        if (sl) atomicAdd(&wide_cnt[(sl - 1u) * kPairRadix + (de >> 8)], c);
    };
    count_keys(add_wide, std::false_type());
    __syncthreads();
    if (tid == 0) hdr[0] = nw;
    if (tid < nw) hdr[1 + tid] = wide_d[tid];
    for (uint32_t k = 0; k < nw; k++)
    {
        const uint32_t d = wide_d[k];
        if (tid < kPairRadix) hdr[16 + k * kPairRadix + tid] = wide_cnt[k * kPairRadix + tid] >> 16;
        if (tid < kPairRowWords) // the T2 row with the low halves of the exact counts (a wrapped even counter had carried into its odd neighbour)
            t2[((size_t) d * nb + b) * kPairRowWords + tid] =
                (wide_cnt[k * kPairRadix + 2 * tid] & 0xFFFFu) | (wide_cnt[k * kPairRadix + 2 * tid + 1] << 16);
    }
}

// ---- what the unit-sum kernels of both digit widths share ----------------------------------------------------------------

// base[d] = first element of digit value d's units (exclusive scan of the leader's digit totals), base[RADIX] = n.
// All 1024 threads call; RADIX <= 256.
template<int RADIX>
__device__ __forceinline__ void pair_unit_bases(const uint32_t* __restrict__ totals_l, uint32_t n, uint32_t* base, uint32_t* tmp)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t t = tid < (uint32_t) RADIX ? totals_l[tid] : 0u;
    uint32_t wtotal;
    uint32_t excl = wave_exclusive_sum(t, lane, wtotal);
    if (lane == 0) tmp[wave] = wtotal;
    __syncthreads();
    excl += sum_of_preceding_waves(tmp, 4, wave < 4 ? wave : 0, lane);
    if (tid < (uint32_t) RADIX) base[tid] = excl;
    if (tid == 0) base[RADIX] = n;
    __syncthreads();
}

// The run of units of follower workgroup w of nb: edge[0] = its first unit, edge[1] = the first unit of the next
// workgroup, edge_start[] = the elements they start at.  Units are numbered u = d * row_len + i (digit value d of the
// leader, i-th block or sub-block), start at base[d] + scanned[d * row_len + i], and starts ascend with u.  The first unit
// of workgroup w is the first one that starts at or after element P = n * w / nb: U(P) = all units of the digit values
// below d* (the last one whose units start before P) plus those of d* whose scanned entry is below P - base[d*].
template<int RADIX>
__device__ __forceinline__ void pair_find_run(const uint32_t* __restrict__ scanned, uint32_t row_len, const uint32_t* base,
                                              uint32_t* found, uint32_t n, uint32_t w, uint32_t nb, uint32_t (&edge)[2],
                                              uint32_t (&edge_start)[2])
{
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (int side = 0; side < 2; side++)
    {
        const uint32_t P = (uint32_t) ((uint64_t) n * (w + side) / nb);
        if (w + side == nb) // (block-uniform branches throughout)
        {
            edge[side] = (uint32_t) RADIX * row_len;
            edge_start[side] = n;
            continue;
        }
        if (tid < 2) found[tid] = 0;
        __syncthreads();
        if (tid < (uint32_t) RADIX && base[tid] < P) atomicAdd(&found[0], 1u); // d* + 1
        __syncthreads();
        const uint32_t below = found[0];
        if (below == 0) // P == 0
        {
            edge[side] = 0;
            edge_start[side] = 0;
            continue;
        }
        const uint32_t ds = below - 1;
        uint32_t mine = 0;
        for (uint32_t i = tid; i < row_len; i += blockDim.x)
            if (base[ds] + scanned[(size_t) ds * row_len + i] < P) mine++;
        if (mine) atomicAdd(&found[1], mine);
        __syncthreads();
        const uint32_t in_row = found[1]; // >= 1: the row's first unit starts at base[ds] < P
        edge[side] = ds * row_len + in_row;
        // the first unit not before P; past the row's end it is the next digit value's first unit
        edge_start[side] = in_row < row_len ? base[ds] + scanned[(size_t) ds * row_len + in_row] : base[ds + 1];
        __syncthreads();
    }
}

// Follower pass `pass` (8-bit digits): workgroup w's run of units and its digit counts.  table_l / totals_l: the leader's
// scanned table and digit totals; table_f: the follower's count table (as radix_count_kernel would write it); ranges[w] =
// the element range of workgroup w.  Returns at once when the follower counts for itself.
static __global__ __launch_bounds__(1024) void radix_pair_unitsum_kernel(const uint32_t* __restrict__ t2,
                                                                  const uint32_t* __restrict__ table_l,
                                                                  const uint32_t* __restrict__ totals_l,
                                                                  uint32_t* __restrict__ table_f, uint2* __restrict__ ranges,
                                                                  uint32_t n, PassPlan* plan, uint32_t pass,
                                                                  uint32_t shift = 0, uint32_t mask = 0, uint32_t plan_flags = 0)
{
    if (plan->off[pass]) // a pass of the sequence not taken (radix_lds_finish.hpp)
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) plan->skip[pass] = kSkipWithoutCounting;
        return;
    }
    if (plan->pair_fallback[pass]) return; // (kernel-uniform)
    shift -= plan->shift_down[pass]; // (PassPlan::top_bit)
    if ((plan_flags & kPlanShortcut) && plan_digit_is_constant(plan, shift, mask)) // an identity, known without any table
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) plan->skip[pass] = kSkipWithoutCounting;
        return;
    }
    __shared__ uint32_t base[kPairRadix + 1];
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t found[2];
    __shared__ uint32_t part[32][kPairRadix];
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = gridDim.x, w = blockIdx.x;
    if (w == 0 && tid == 0) plan->skip[pass] = 0;
    pair_unit_bases<kPairRadix>(totals_l, n, base, tmp);
    uint32_t edge[2], edge_start[2];
    pair_find_run<kPairRadix>(table_l, nb, base, found, n, w, nb, edge, edge_start);

    // a run of very many (tiny) units would be a long chain of table reads in this one workgroup: count instead
    if (edge[1] - edge[0] > kPairMaxRunUnits)
    {
        if (tid == 0) plan->pair_fallback[pass] = 1;
        return;
    }
    // digit counts of the run: sum of its T2 rows; 32 rows at a time, thread (g, q) adds words 4q .. 4q + 3 of the rows
    // g, g + 32, ...
    const uint32_t g = tid >> 5, q = tid & 31u;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* rows = reinterpret_cast<const uint4*>(t2);
#pragma unroll 4
    for (uint32_t u = edge[0] + g; u < edge[1]; u += 32)
    {
        const uint4 v = rows[(size_t) u * (kPairRowWords / 4) + q];
        acc[0] += v.x & 0xFFFFu, acc[1] += v.x >> 16, acc[2] += v.y & 0xFFFFu, acc[3] += v.y >> 16;
        acc[4] += v.z & 0xFFFFu, acc[5] += v.z >> 16, acc[6] += v.w & 0xFFFFu, acc[7] += v.w >> 16;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) part[g][8 * q + k] = acc[k];
    __syncthreads();
    if (tid < kPairRadix)
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 32; k++) c += part[k][tid];
        table_f[(size_t) tid * nb + w] = c;
    }
    if (tid == 0) ranges[w] = make_uint2(edge_start[0], edge_start[1]);
}

// ---- 4-bit digits (the reference's pass structure) -------------------------------------------------------------------------
// 16 digit values x nb blocks are too few units to balance runs on (a unit would be 1/16 of a workgroup's share), so a
// leader's block is cut into kPairSub sub-blocks (whole tiles, in order) and the units are (digit value, sub-block): as many
// and as small as with 8-bit digits.  The two-digit histogram of a sub-block is 16 x 16 32-bit counters -- nothing overflows,
// and no unit is longer than 1/16 of a share, so the only way back to counting is a run of very many tiny units.
constexpr uint32_t kPairSub = 16;
constexpr uint32_t kPair4Radix = 16;

// Leader's count kernel, 4-bit digits: one 256-thread workgroup per SUB-BLOCK (grid = nb * kPairSub; sub-block j of block b
// = the j-th sixteenth of block b's tiles), so that no workgroup drains its loads sixteen times.  table_sub: [16][nb *
// kPairSub] counts per (digit value, sub-block); the row-scan kernel behind this one scans it and copies every kPairSub-th
// entry into the leader's usual [16][nb] table (the scanned count of a block = that of its first sub-block).  t2: per
// sub-block 256 words, word d * 16 + e = #keys with digit p = d and digit p + 1 = e.
template<typename KeyT, int TILE, bool XF = false, bool COLLECT = false>
__global__ __launch_bounds__(256) void radix_pair4_count_kernel(const KeyT* __restrict__ keys_a, uint32_t* __restrict__ table_sub,
                                                                uint32_t* __restrict__ t2, uint32_t n, uint32_t shift,
                                                                uint32_t mask, uint32_t shift2, uint32_t mask2,
                                                                uint32_t tiles_total, uint32_t xform, const KeyT* keys_b,
                                                                PassPlan* plan, uint32_t pass, uint32_t plan_flags = 0,
                                                                uint32_t share = 0)
{
    constexpr int THREADS = 256;
    constexpr int WAVES = THREADS / kWave;
    __shared__ uint32_t hist[WAVES][256]; // wave-private counters of the combined digit d | e << 4
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = gridDim.x / kPairSub, b = blockIdx.x / kPairSub, j = blockIdx.x % kPairSub;
    const KeyT* __restrict__ keys = keys_a;
    if (plan)
    {
        if ((plan_flags & kPlanShortcut) && plan_digit_is_constant(plan, shift, mask)) // (as in radix_pair_count_kernel)
        {
            if (blockIdx.x == 0 && tid == 0)
            {
                plan->skip[pass] = kSkipWithoutCounting;
                plan->pair_fallback[pass + 1] = 1;
            }
            return;
        }
        if (pass > 0 && plan->flip[pass]) keys = keys_b;
        if (blockIdx.x == 0 && tid == 0)
        {
            plan->skip[pass] = 0;
            if (COLLECT && (plan_flags & kPlanCollectBits)) plan->bits_valid = 1;
        }
    }
    KeyT acc_or = 0, acc_and = (KeyT) ~(KeyT) 0; // of the raw keys this thread reads
    for (int i = tid; i < WAVES * 256; i += THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    // sub-block j of block b: the j-th sixteenth of the block's tiles, or (share > 0: blocks are equal shares of the
    // elements) the j-th piece of ceil(share / 16) elements rounded up to 64
    uint64_t begin, end;
    if (share)
    {
        uint64_t b0, b1;
        block_range(b, nb, tiles_total, TILE, n, share, b0, b1);
        const uint32_t sub = ((share + kPairSub - 1) / kPairSub + 63u) & ~63u;
        begin = b0 + (uint64_t) j * sub;
        if (begin > b1) begin = b1;
        end = begin + sub;
        if (end > b1) end = b1;
    }
    else
    {
        uint32_t first, last, sf, sl;
        block_tile_range(b, nb, tiles_total, first, last);
        block_tile_range(j, kPairSub, last - first, sf, sl);
        begin = (uint64_t) (first + sf) * TILE;
        end = (uint64_t) (first + sl) * TILE;
        if (end > n) end = n;
    }
    uint32_t* my_hist = hist[wave];
    const KeyCodec<KeyT, XF> codec_in(xform & 3u);
    auto combined = [&](KeyT raw) {
        const KeyT k = codec_in.encode(raw);
        return digit_of<KeyT>(k, shift, mask) | (digit_of<KeyT>(k, shift2, mask2) << 4);
    };
    // every lane of the wave is active when this runs
    TallyRun run;
    auto add_count = [&](uint32_t cv, uint32_t c) { atomicAdd(&my_hist[cv], c); };
    auto tally = [&](auto peel, KeyT raw) { wave_tally_mode(peel, combined(raw), lane, run, add_count); };
    constexpr int VEC = 16 / sizeof(KeyT);
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    auto tally_vec = [&](auto peel, const VecT& a) {
        if constexpr (sizeof(KeyT) == 4)
        {
            if (COLLECT) acc_or |= a.x | a.y | a.z | a.w, acc_and &= a.x & a.y & a.z & a.w;
            tally(peel, a.x); tally(peel, a.y); tally(peel, a.z); tally(peel, a.w);
        }
        else
        {
            if (COLLECT) acc_or |= a.x | a.y, acc_and &= a.x & a.y;
            tally(peel, a.x); tally(peel, a.y);
        }
    };
    if (begin < end) // (block-uniform)
    {
        const uint64_t nvec = vec_ok ? (end - begin) / VEC : 0;
        const VecT* vkeys = reinterpret_cast<const VecT*>(keys + begin);
        uint64_t vbase = 0;
        auto main_loop = [&](auto peel) { // (twice, chosen once per wave from its first keys: see wave_tally)
            for (; vbase + 4 * THREADS <= nvec; vbase += 4 * THREADS) // block-uniform trip count, 4 x 16 B in flight per lane
            {
                if (wave_tally_gives_up(peel, run)) break; // (peel mode only; the wave comes back in the stateless mode)
                VecT a = load_streaming(&vkeys[vbase + tid]);
                VecT bq = load_streaming(&vkeys[vbase + tid + THREADS]);
                VecT c = load_streaming(&vkeys[vbase + tid + 2 * THREADS]);
                VecT d = load_streaming(&vkeys[vbase + tid + 3 * THREADS]);
                tally_vec(peel, a);
                tally_vec(peel, bq);
                tally_vec(peel, c);
                tally_vec(peel, d);
            }
        };
        if (4 * THREADS <= nvec) wave_tally_dispatch(combined(vkeys[tid].x), lane, run, main_loop, add_count);
        for (uint64_t i = begin + vbase * VEC + tid; i < end; i += THREADS)
        {
            const KeyT raw = keys[i];
            if (COLLECT) acc_or |= raw, acc_and &= raw;
            atomicAdd(&my_hist[combined(raw)], 1u);
        }
    }
    __syncthreads();
    if (COLLECT && (plan_flags & kPlanCollectBits)) plan_publish_bits<KeyT>(plan, acc_or, acc_and, lane);
    // thread = combined value: d = tid & 15, e = tid >> 4
    uint32_t c = 0;
#pragma unroll
    for (int wv = 0; wv < WAVES; wv++) c += hist[wv][tid];
    t2[(size_t) blockIdx.x * 256 + (tid & 15u) * 16 + (tid >> 4)] = c;
    // the sub-block's count of digit value d = sum over e: the lanes d, d + 16, d + 32, d + 48 of the four waves
    uint32_t sum = c;
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    __syncthreads(); // every thread has read its column of hist
    if (lane < 16) hist[wave][lane] = sum;
    __syncthreads();
    if (tid < kPair4Radix)
        table_sub[(size_t) tid * gridDim.x + blockIdx.x] = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
}

// Follower pass `pass` (4-bit digits): as radix_pair_unitsum_kernel, units = (digit value, sub-block).
// sub_scanned: the leader's per-sub-block table after its row scan; totals_l: the leader's digit totals.
static __global__ __launch_bounds__(1024) void radix_pair4_unitsum_kernel(const uint32_t* __restrict__ t2,
                                                                   const uint32_t* __restrict__ sub_scanned,
                                                                   const uint32_t* __restrict__ totals_l,
                                                                   uint32_t* __restrict__ table_f, uint2* __restrict__ ranges,
                                                                   uint32_t n, PassPlan* plan, uint32_t pass,
                                                                   uint32_t shift = 0, uint32_t mask = 0, uint32_t plan_flags = 0)
{
    if (plan->pair_fallback[pass]) return; // (kernel-uniform)
    if ((plan_flags & kPlanShortcut) && plan_digit_is_constant(plan, shift, mask)) // an identity, known without any table
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) plan->skip[pass] = kSkipWithoutCounting;
        return;
    }
    __shared__ uint32_t base[kPair4Radix + 1];
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t found[2];
    __shared__ uint32_t part[64][kPair4Radix];
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = gridDim.x, w = blockIdx.x;
    const uint32_t row_len = nb * kPairSub;
    if (w == 0 && tid == 0) plan->skip[pass] = 0;
    pair_unit_bases<kPair4Radix>(totals_l, n, base, tmp);
    uint32_t edge[2], edge_start[2];
    pair_find_run<kPair4Radix>(sub_scanned, row_len, base, found, n, w, nb, edge, edge_start);
    if (edge[1] - edge[0] > kPairMaxRunUnits * 4) // (rows of 64 bytes here: four times as many cost the same)
    {
        if (tid == 0) plan->pair_fallback[pass] = 1;
        return;
    }
    // unit u = d * row_len + i: the 16 words d * 16 .. d * 16 + 15 of sub-block i's T2 row; thread (g, e) adds word e of the
    // units g, g + 64, ...
    const uint32_t g = tid >> 4, e = tid & 15u;
    uint32_t acc = 0;
#pragma unroll 4
    for (uint32_t u = edge[0] + g; u < edge[1]; u += 64)
    {
        const uint32_t d = u / row_len, i = u % row_len;
        acc += t2[(size_t) i * 256 + d * 16 + e];
    }
    part[g][e] = acc;
    __syncthreads();
    if (tid < kPair4Radix)
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 64; k++) c += part[k][tid];
        table_f[(size_t) tid * nb + w] = c;
    }
    if (tid == 0) ranges[w] = make_uint2(edge_start[0], edge_start[1]);
}

} // namespace glu_hip
