// radix_sort_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the LSD radix sort.
//
// Replaces the device side of glu::RadixSort (reference glu/RadixSort.hpp):
//   K1  radix_count_kernel    <- k_radix_sort_counting_shader   (RadixSort.hpp:11-58)
//   K2  radix_row_scan_kernel <- BlellochScan upsweep/downsweep over the [radix][block] table
//                                (BlellochScan.hpp:13-76, called at RadixSort.hpp:311)
//   K4  radix_scatter_kernel  <- k_radix_sort_reordering_shader (RadixSort.hpp:60-183)
// Same contract (stable counting pass on one digit), different machine mapping:
//   * a fixed grid of persistent workgroups, each owning a contiguous range of tiles, so the digit table is
//     [RADIX][num_blocks] with num_blocks <= a few thousand instead of 16 x N/1024;
//   * per-workgroup digit counts live in registers/LDS, one plain store per (digit, block) -- no global atomics;
//   * stable local ranks from wave64 ballots (match-any on the digit bits) + mbcnt, wave-private LDS counters;
//   * (key, val) staged through LDS in ranked order so that every digit leaves as one contiguous burst.
// The digit width BITS is a template parameter (4 = the reference's pass structure, 8 = half the passes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "device_utils.hpp"

#ifndef GLU_DMA_AUX
#define GLU_DMA_AUX 0 // cache-policy bits of the experimental LDS-DMA tile loads (tuning builds)
#endif

namespace glu_hip
{
constexpr int kWave = 64;

// Digit of a key: bits [shift, shift + popcount(mask)).  `mask` is always a run of low one-bits (a narrower last digit
// reuses the wider kernel), so this is one v_bfe_u32 instead of a shift and an and; for 64-bit keys the word holding the
// digit is picked first (digits of 4 / 8 bits at multiples of their width never straddle the two words).
template<typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, uint32_t shift, uint32_t mask)
{
    const uint32_t width = (uint32_t) __popc(mask);
    if constexpr (sizeof(KeyT) == 4)
        return __builtin_amdgcn_ubfe((uint32_t) key, shift, width);
    else
    {
        // branch-free word pick (a uniform branch here would split the unrolled per-item loops into basic blocks)
        const uint32_t hi_sel = shift >= 32u ? 0xFFFFFFFFu : 0u;
        const uint32_t word = ((uint32_t) key & ~hi_sel) | ((uint32_t) (key >> 32) & hi_sel);
        return __builtin_amdgcn_ubfe(word, shift & 31u, width);
    }
}

// Key encodings.  The kernels sort unsigned bit patterns; signed-integer and IEEE-float keys are mapped to unsigned
// patterns with the same order on the first pass's loads and mapped back on the last pass's stores (kernel-uniform
// switches, nothing extra touches memory).  xform = input transform | output transform << 2.
enum KeyTransform : uint32_t
{
    KEY_XF_NONE = 0,
    KEY_XF_SIGNED = 1, // flip the sign bit
    KEY_XF_FLOAT = 2   // negative: flip every bit, non-negative: flip the sign bit (total order, -0 < +0, NaNs at the ends)
};

// Branch-free form (a uniform `if` per key would split the unrolled load / store loops into basic blocks and cost
// the scatter kernel 50 %): code(k) = k ^ ((sign_fill(k or ~k) & all) | sign), with (all, sign) = (0, 0) for no
// transform, (0, SIGN) for signed integers, (~0, SIGN) for floats.
template<typename KeyT, bool XF = true>
struct KeyCodec
{
    using S = typename std::make_signed<KeyT>::type;
    static constexpr int TOP = sizeof(KeyT) * 8 - 1;
    KeyT all, sign;
    __device__ __forceinline__ explicit KeyCodec(uint32_t xf) :
        all(xf == KEY_XF_FLOAT ? (KeyT) ~(KeyT) 0 : (KeyT) 0),
        sign(xf == KEY_XF_NONE ? (KeyT) 0 : (KeyT) 1 << TOP)
    {
    }
    __device__ __forceinline__ KeyT encode(KeyT k) const { return k ^ (((KeyT) ((S) k >> TOP) & all) | sign); }
    __device__ __forceinline__ KeyT decode(KeyT k) const { return k ^ (((KeyT) ((S) ~k >> TOP) & all) | sign); }
};
// XF = false: the kernels instantiated for plain unsigned keys and for the middle passes of typed sorts carry no codec
// arithmetic at all (3 VALU instructions per 32-bit key on each side, twice that for 64-bit keys, in kernels whose
// per-tile time is set by VALU issue as much as by memory).
template<typename KeyT>
struct KeyCodec<KeyT, false>
{
    __device__ __forceinline__ explicit KeyCodec(uint32_t) {}
    __device__ __forceinline__ KeyT encode(KeyT k) const { return k; }
    __device__ __forceinline__ KeyT decode(KeyT k) const { return k; }
};

// Contiguous tile range [first, last) owned by workgroup `b` of `nb`, tiles_total >= nb.
__device__ __forceinline__ void block_tile_range(uint32_t b, uint32_t nb, uint32_t tiles_total, uint32_t& first,
                                                 uint32_t& last)
{
    uint32_t q = tiles_total / nb, r = tiles_total % nb;
    first = b * q + (b < r ? b : r);
    last = first + q + (b < r ? 1u : 0u);
}

// Wave64 exclusive prefix sum with DPP moves (VALU only; the __shfl_up form goes through the LDS crossbar with
// ds_bpermute, 6 dependent LDS round trips).  update_dpp(0, x, ...) yields 0 in lanes without a source lane / in rows
// outside row_mask, so no lane predicate is needed.
__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v, uint32_t lane, uint32_t& total)
{
    (void) lane;
    int incl = (int) v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false); // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false); // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false); // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false); // row_shr:8   -> scanned inside rows of 16
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    total = (uint32_t) __builtin_amdgcn_readlane(incl, kWave - 1);
    return (uint32_t) incl - v;
}

// Element range [begin, end) of workgroup b.  share == 0: its whole tiles (block_tile_range; the kernels of the small
// geometry, whose scatter works on whole tiles).  share > 0 (the line kernel and the count kernels in front of it): an
// equal share of the elements, a multiple of 64 -- no workgroup gets a whole tile more than another, which is what made
// sort time a sawtooth over n between 4 M and 60 M elements (2.3 tiles per workgroup = some with 3, the rest waiting).
__device__ __forceinline__ void block_range(uint32_t b, uint32_t nb, uint32_t tiles_total, uint32_t tile, uint32_t n,
                                            uint32_t share, uint64_t& begin, uint64_t& end)
{
    if (share)
    {
        begin = (uint64_t) b * share;
        if (begin > n) begin = n;
        end = begin + share;
    }
    else
    {
        uint32_t first, last;
        block_tile_range(b, nb, tiles_total, first, last);
        begin = (uint64_t) first * tile;
        end = (uint64_t) last * tile;
    }
    if (end > n) end = n;
}

// Sum of the first `wave` entries of an LDS array of per-wave totals (count <= 64), without a branch per entry: lane l
// reads entry l, the wave scans them, and the (wave-uniform) result is read from lane `wave`.
__device__ __forceinline__ uint32_t sum_of_preceding_waves(const uint32_t* totals, int count, uint32_t wave, uint32_t lane)
{
    const uint32_t v = lane < (uint32_t) count ? totals[lane] : 0u;
    uint32_t all;
    const uint32_t excl = wave_exclusive_sum(v, lane, all);
    return (uint32_t) __builtin_amdgcn_readlane((int) excl, (int) __builtin_amdgcn_readfirstlane((int) wave));
}

// ---------------------------------------------------------------------------------------------------------
// Pass plan (device memory, one per sort object; only the sort of >= kPlanMinCount elements uses it).  A counting pass
// whose digit has the same value in every key is an identity permutation; its scatter is skipped.  Which pair of arrays
// holds the data before pass p then depends on the data, so it is tracked on the device:
//   flip[p]  1 = the data is in the scratch arrays ("B") before pass p, 0 = in the caller's ("A"); flip[0] = 0
//   skip[p]  set by the row-scan kernel of pass p when one digit value holds all n keys, reset by its count kernel
// The scatter kernel of pass p publishes flip[p + 1] = flip[p] ^ !skip[p]; the count kernel of pass p + 1 runs after it.
// ---------------------------------------------------------------------------------------------------------
constexpr int kPlanMaxPasses = 32;
// (The part zeroed at the start of every sort begins and ends on a 64-byte boundary: hipMemsetAsync of a range that does not
// becomes three fill kernels -- head, body, tail -- 14 us in front of every planned sort instead of 4.5.)
struct alignas(64) PassPlan
{
    uint32_t flip[kPlanMaxPasses + 1];
    uint32_t skip[kPlanMaxPasses];
    // paired passes (radix_pair_passes.hpp): 1 = pass p counts for itself although its table was to come from the
    // two-digit histogram of pass p - 1; zeroed at the start of every sort, set by kernels that run before pass p
    alignas(64) uint32_t pair_fallback[kPlanMaxPasses + 1];
    // A sort that tries to end in LDS (radix_lds_finish.hpp) enqueues two alternative sequences of passes; off[p] = 1: pass
    // p belongs to the sequence not taken -- its count kernels return at once and report it as skipped without counting.
    // finish = 1: the top-bit passes ran and the last pass orders every run in LDS; finish_longest: the longest run seen
    // (~0: never counted).  Zeroed with pair_fallback, set by radix_finish_plan_kernel.
    uint32_t off[kPlanMaxPasses];
    uint32_t finish, finish_longest;
    // (round 6) finish_rounds = 1: every run of the in-LDS pass is ordered by ballot rounds (radix_finish_sort_kernel) -- fewer than
    // nine key bits are left to order or vary there, which one round does faster than radix_finish_bucket_kernel's buckets fill
    uint32_t finish_rounds;
    // Which key bits vary over the input: collected by the count kernel of the first pass of an untyped sort (the OR of
    // all keys and the OR of all complemented keys, low / high word).  A later pass whose digit lies in bits that do not
    // vary is an identity: its count kernel says so (skip[p] = 1) without reading the keys.  Zeroed with pair_fallback.
    uint32_t bits_valid;
    uint32_t bits_or[2], bits_nor[2];
    // A sort that tries to end in LDS takes its runs from the 16 key bits below top_bit, which the DEVICE chooses (round 5,
    // radix_sample_top_kernel: the highest bit that varies in a sample of the keys -- small-range keys end in LDS on an object's
    // first sort).  The host enqueues the two top-bit passes for the whole key's top bits; their kernels shift down by
    // shift_down[pass] = key bits - top_bit.  0: the host's shifts stand.  Zeroed with pair_fallback.
    uint32_t top_bit;
    uint32_t shift_down[kPlanMaxPasses];
    // (radix_sample_top_kernel's workgroups: OR of the sampled keys, OR of their complements, low / high word; workgroups done)
    uint32_t sample_or[2], sample_nor[2], sample_done;
};

// PassPlan::skip values: 0 = the pass runs, 1 = an identity found by the row scan (one digit value holds every key; the
// tables of the pass are complete), kSkipWithoutCounting = an identity known before counting (no tables)
constexpr uint32_t kSkipWithoutCounting = 2;

// flags of the count kernels of planned sorts
constexpr uint32_t kPlanCollectBits = 1; // first pass of an untyped sort: collect which key bits vary
constexpr uint32_t kPlanShortcut = 2;    // a later pass that may be skipped: return at once if its digit cannot vary

__device__ __forceinline__ bool plan_digit_is_constant(const PassPlan* plan, uint32_t shift, uint32_t mask)
{
    if (!plan->bits_valid) return false;
    const uint32_t w = shift >> 5; // a digit never straddles the two words of a 64-bit key
    return (((plan->bits_or[w] & plan->bits_nor[w]) >> (shift & 31u)) & mask) == 0;
}

// One counter update per key for a whole wave: add(value, count) is called by every lane that must add `count` to the
// counter of `value`.  Every lane of the wave is active when this runs.
//   * all 64 values equal (constant keys, sorted input): one add of 64 by lane 0;
//   * PEEL (the caller saw many equal values in this stretch of the input, wave_many_equal): few distinct keys in any
//     order make 32-way same-address LDS atomics, 32 turns each, and the pair kernel's shared counters are hit by all 16
//     waves at once.  Groups of equal values are peeled off one by one, a ballot and one add by the group's first lane
//     each, for as long as they hold at least kTallyPeelMin lanes; whoever is left adds for himself.
//   * else every lane adds 1.
// 2^26 pairs with two distinct key values in random order: first pair-count kernel 276 -> 156 us (tools/trace_one_sort.sh).
// The choice is made once per wave (from the first keys of its share of the input) between two copies of the count loop:
// a test per key cost uniform keys 10 % of the pair-count kernel, one per loop iteration 4.5 % (same-box A/B, tools/ab_lib.sh).
constexpr int kTallyPeelMin = 8, kTallyPeelRounds = 8;
__device__ __forceinline__ bool wave_many_equal(uint32_t v) // wave-uniform
{
    const uint64_t same = __ballot(v == (uint32_t) __builtin_amdgcn_readfirstlane(v));
    return same != ~0ull && __popcll(same) >= kTallyPeelMin;
}
template<bool PEEL, typename F>
__device__ __forceinline__ void wave_tally(uint32_t v, uint32_t lane, F&& add)
{
    const uint32_t v0 = __builtin_amdgcn_readfirstlane(v);
    if constexpr (!PEEL)
    {
        if (__ballot(v != v0) == 0)
        {
            // (the empty asm keeps the compiler from merging the two branches into one predicated body with selects: 2.5 %
            // of the pair-count kernel on uniform keys)
            asm volatile("");
            if (lane == 0) add(v0, 64u);
        }
        else
            add(v, 1u);
    }
    else
    {
        const uint64_t same = __ballot(v == v0);
        if (same == ~0ull)
        {
            if (lane == 0) add(v0, 64u);
            return;
        }
        uint64_t todo = ~0ull, grp = same;
        uint32_t vg = v0, first = 0;
        for (int round = 0; __popcll(grp) >= kTallyPeelMin;)
        {
            if (lane == first) add(vg, (uint32_t) __popcll(grp));
            todo &= ~grp;
            if (todo == 0 || ++round == kTallyPeelRounds) break;
            first = (uint32_t) __ffsll((unsigned long long) todo) - 1u; // (wave-uniform: todo is made of ballots)
            vg = (uint32_t) __builtin_amdgcn_readlane((int) v, (int) first);
            grp = __ballot(v == vg); // (a group smaller than kTallyPeelMin ends the loop: it and the rest add one by one)
        }
        if ((todo >> lane) & 1ull) add(v, 1u);
    }
}

// Third copy of the count loop, for waves whose first 64 keys are all equal (constant keys -- the reference README's
// benchmark input --, sorted or run-structured input): consecutive all-equal steps with the same value only grow a
// wave-uniform run length; the counters are touched when the value changes (and every 2^15 keys, so that one add never
// exceeds what a 16-bit counter of the pair kernel can take without its overflow being seen).  2^28 all-zero keys: the
// whole sort (one read of the keys and fourteen launches that return at once) 0.342 -> 0.318 ms in a same-box A/B
// (tools/ab_zero.sh); uniform keys unchanged.  A step that is not all-equal takes the peeling path.
struct TallyRun
{
    uint32_t value = 0, count = 0; // wave-uniform
    // (peel mode, wave_tally_cached) the values this wave keeps counting in registers, and how many of each it has seen since their
    // last flush; ~0 is no digit value of any kernel
    static constexpr int kCached = 8;
    uint32_t hot_value[kCached] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, hot_count[kCached] = {0, 0, 0, 0, 0, 0, 0, 0}; // wave-uniform
    uint32_t steps = 0, misses = 0, gave_up = 0; // (wave_tally_gives_up: more frequent values than registers, the stateless peel takes over)
};
template<typename F>
__device__ __forceinline__ void wave_tally_flush(TallyRun& run, uint32_t lane, F&& add)
{
    if (run.count != 0 && lane == 0) add(run.value, run.count);
    run.count = 0;
}
template<typename F>
__device__ __forceinline__ void wave_tally_runs(uint32_t v, uint32_t lane, TallyRun& run, F&& add)
{
    const uint32_t v0 = __builtin_amdgcn_readfirstlane(v);
    if (__ballot(v != v0) == 0)
    {
        if (v0 != run.value || run.count >= (1u << 15))
        {
            wave_tally_flush(run, lane, add);
            run.value = v0;
        }
        run.count += 64u;
    }
    else
        wave_tally<true>(v, lane, add);
}
// Peel mode with memory (round 5): a handful of key values in random order -- flags, categories, three values -- made the
// stateless peel above the whole cost of a count kernel (three values, 2^28 pairs: the pair-count kernel 0.80 ms against 0.20 ms
// on uniform keys: three ballots, three pairs of one-lane atomics and their bookkeeping per 64 keys; now 0.24 ms).  Here the wave
// counts up to eight values in (wave-uniform) registers: a compare, a ballot and an add each per step, no counter is touched
// while the step's values are all among them.  A step with other values gives the first of them the place of the entry seen
// least since its flush (which is added to the counters then) and lets the lanes left over add one by one.
// wave_tally_cached_flush adds what is still in the registers.  Inputs whose frequent values keep changing -- more than eight of
// them, or short runs of equal keys one after the other (the later passes of a sort of Zipf-distributed keys) -- make every other
// step such a step: a wave that sees that (wave_tally_gives_up) leaves its loop, flushes and goes on in the stateless peel mode,
// which is what such inputs need.  Counts are not bounded: the pair kernels' 16-bit counters overflow for such inputs anyway, and an
// overflow -- by many small adds or one large one -- breaks the row sums that radix_pair_count_kernel checks against the exact
// 32-bit table; the 32-bit counters cannot overflow (a count is at most the input size).
template<typename F>
__device__ __forceinline__ void wave_tally_cached(uint32_t v, uint32_t lane, TallyRun& st, F&& add)
{
    uint64_t todo = ~0ull;
#pragma unroll
    for (int k = 0; k < TallyRun::kCached; k++)
    {
        const uint64_t m = __ballot(v == st.hot_value[k]);
        st.hot_count[k] += (uint32_t) __popcll(m);
        todo &= ~m;
    }
    if (todo != 0) // (wave-uniform)
    {
        st.misses++;
        int victim = 0;
        uint32_t fewest = st.hot_count[0]; // (no dynamic indexing of the register arrays: they would live in scratch memory)
#pragma unroll
        for (int k = 1; k < TallyRun::kCached; k++)
            if (st.hot_count[k] < fewest) fewest = st.hot_count[k], victim = k;
        const uint32_t first = (uint32_t) __ffsll((unsigned long long) todo) - 1u;
        const uint32_t vg = (uint32_t) __builtin_amdgcn_readlane((int) v, (int) first);
        const uint64_t grp = __ballot(v == vg);
#pragma unroll
        for (int k = 0; k < TallyRun::kCached; k++)
            if (k == victim)
            {
                if (st.hot_count[k] != 0 && lane == 0) add(st.hot_value[k], st.hot_count[k]);
                st.hot_value[k] = vg;
                st.hot_count[k] = (uint32_t) __popcll(grp);
            }
        todo &= ~grp;
        if ((todo >> lane) & 1ull) add(v, 1u);
    }
}
// called once per iteration of a count loop (16 keys per lane): true = this wave's loop in peel mode should stop -- more than
// half of the last 32 steps met values the registers did not hold (filling them takes eight such steps at most) -- and the wave
// go on in the stateless mode (wave-uniform)
template<typename Mode>
__device__ __forceinline__ bool wave_tally_gives_up(Mode, TallyRun& st)
{
    if constexpr (Mode::value != 1)
        return false;
    else
    {
        if ((++st.steps & 1u) == 0)
        {
            if (st.misses > 16u) st.gave_up = 1;
            st.misses = 0;
        }
        return st.gave_up != 0;
    }
}
template<typename F>
__device__ __forceinline__ void wave_tally_cached_flush(TallyRun& st, uint32_t lane, F&& add)
{
#pragma unroll
    for (int k = 0; k < TallyRun::kCached; k++)
    {
        if (st.hot_count[k] != 0 && lane == 0) add(st.hot_value[k], st.hot_count[k]);
        st.hot_count[k] = 0;
    }
}
__device__ __forceinline__ bool wave_all_equal(uint32_t v) // wave-uniform
{
    return __ballot(v != (uint32_t) __builtin_amdgcn_readfirstlane(v)) == 0;
}
// mode tags of the count loops: 0 = plain, 1 = peel groups of equal values, 2 = run lengths
using TallyPlain = std::integral_constant<int, 0>;
using TallyPeel = std::integral_constant<int, 1>;
using TallyRuns = std::integral_constant<int, 2>;
using TallyStateless = std::integral_constant<int, 3>; // the peel without memory: what a wave falls back to from mode 1
template<typename Mode, typename F>
__device__ __forceinline__ void wave_tally_mode(Mode, uint32_t v, uint32_t lane, TallyRun& run, F&& add)
{
    if constexpr (Mode::value == 2)
        wave_tally_runs(v, lane, run, add);
    else if constexpr (Mode::value == 1)
        wave_tally_cached(v, lane, run, add);
    else
        wave_tally<Mode::value == 3>(v, lane, add);
}
// runs `loop(mode)` in the mode the wave's first value suggests
template<typename L, typename F>
__device__ __forceinline__ void wave_tally_dispatch(uint32_t first_value, uint32_t lane, TallyRun& run, L&& loop, F&& add)
{
    if (wave_all_equal(first_value))
    {
        loop(TallyRuns());
        wave_tally_flush(run, lane, add);
    }
    else if (wave_many_equal(first_value))
    {
        run.steps = 0, run.misses = 0, run.gave_up = 0;
        loop(TallyPeel());
        wave_tally_cached_flush(run, lane, add);
        if (run.gave_up) loop(TallyStateless()); // (the rest of the wave's keys)
    }
    else
        loop(TallyPlain());
}

// OR / AND of the keys a thread has seen -> the plan, once per WORKGROUP.  Called by every thread of the workgroup (it
// holds a barrier).  (Once per wave, the first version, cost a 6 M-pair sort 60 of its 250 us: the 4096 waves of a balanced
// launch finish together, every one of them still reads zeros in the plan's words, and their same-address atomics then
// take ~10 ns each, one after the other -- tools/trace_one_sort.sh.)
template<typename KeyT>
__device__ __forceinline__ void plan_publish_bits(PassPlan* plan, KeyT acc_or, KeyT acc_and, uint32_t lane)
{
    __shared__ uint32_t wave_bits[16][4]; // per wave: or lo, nor lo, or hi, nor hi
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        if constexpr (sizeof(KeyT) == 4)
        {
            acc_or |= (KeyT) __shfl_xor((uint32_t) acc_or, o);
            acc_and &= (KeyT) __shfl_xor((uint32_t) acc_and, o);
        }
        else
        {
            acc_or |= (KeyT) __shfl_xor((unsigned long long) acc_or, o);
            acc_and &= (KeyT) __shfl_xor((unsigned long long) acc_and, o);
        }
    }
    const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    if (lane == 0)
    {
        wave_bits[wave][0] = (uint32_t) acc_or;
        wave_bits[wave][1] = ~(uint32_t) acc_and;
        wave_bits[wave][2] = sizeof(KeyT) == 8 ? (uint32_t) ((uint64_t) acc_or >> 32) : 0u;
        wave_bits[wave][3] = sizeof(KeyT) == 8 ? ~(uint32_t) ((uint64_t) acc_and >> 32) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < (sizeof(KeyT) == 8 ? 4u : 2u))
    {
        uint32_t bits = 0;
        for (uint32_t w = 0; w < waves; w++) bits |= wave_bits[w][threadIdx.x];
        // a workgroup that has nothing to add -- nearly all of them when they finish at different times, the words fill up
        // with the first few -- only reads.  (A stale read can only show fewer bits than there are: one atomic too many.)
        uint32_t* word = threadIdx.x == 0 ? &plan->bits_or[0] : threadIdx.x == 1 ? &plan->bits_nor[0] : threadIdx.x == 2 ? &plan->bits_or[1] : &plan->bits_nor[1];
        if ((*reinterpret_cast<volatile uint32_t*>(word) | bits) != *reinterpret_cast<volatile uint32_t*>(word)) atomicOr(word, bits);
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1: per-workgroup digit histogram.  table[d * num_blocks + b] = #keys of block b's range with digit d.
// Reads 1 key per pair (sizeof(KeyT) bytes), writes RADIX counters per workgroup.
// ---------------------------------------------------------------------------------------------------------
template<typename KeyT, int BITS, int THREADS, int TILE, bool XF = false, bool COLLECT = false>
__global__ __launch_bounds__(THREADS) void radix_count_kernel(const KeyT* __restrict__ keys_a,
                                                              uint32_t* __restrict__ table, uint32_t n,
                                                              uint32_t shift, uint32_t mask, uint32_t tiles_total,
                                                              uint32_t xform = 0, const KeyT* keys_b = nullptr,
                                                              PassPlan* plan = nullptr, uint32_t pass = 0,
                                                              bool pair_follower = false, uint32_t plan_flags = 0,
                                                              uint32_t share = 0)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int WAVES = THREADS / kWave;
    const uint32_t MASK = mask; // <= RADIX - 1 (narrower digits reuse the next wider instantiation)
    __shared__ uint32_t hist[WAVES][RADIX];
    // the follower of a pair of passes has its table from the two-digit histogram of the pass before it
    // (radix_pair_passes.hpp) unless a kernel before this one found that it cannot (kernel-uniform)
    if (pair_follower && !plan->pair_fallback[pass]) return;
    if (plan) shift -= plan->shift_down[pass]; // (PassPlan::top_bit)

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // planned sorts: read whichever pair of arrays holds the data before this pass; arm the pass's skip flag
    const KeyT* __restrict__ keys = keys_a;
    if (plan)
    {
        // a pass of the sequence not taken (PassPlan::off), or a digit in key bits that do not vary (the first pass's count
        // kernel collected them): an identity pass
        if (plan->off[pass] || ((plan_flags & kPlanShortcut) && plan_digit_is_constant(plan, shift, MASK))) // (kernel-uniform)
        {
            if (blockIdx.x == 0 && tid == 0) plan->skip[pass] = kSkipWithoutCounting;
            return;
        }
        if (pass > 0 && plan->flip[pass]) keys = keys_b;
        if (blockIdx.x == 0 && tid == 0)
        {
            plan->skip[pass] = 0;
            if (COLLECT && (plan_flags & kPlanCollectBits)) plan->bits_valid = 1;
        }
    }
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();

    uint64_t begin, end;
    block_range(blockIdx.x, gridDim.x, tiles_total, TILE, n, share, begin, end);

    uint32_t* my_hist = hist[wave];
    KeyT acc_or = 0, acc_and = (KeyT) ~(KeyT) 0; // of the raw keys this thread reads

    // every lane of the wave is active when this runs (wave-uniform trip counts below)
    TallyRun run;
    auto add_count = [&](uint32_t dv, uint32_t c) { atomicAdd(&my_hist[dv], c); };
    auto tally = [&](auto peel, uint32_t d) { wave_tally_mode(peel, d, lane, run, add_count); };

    constexpr int VEC = 16 / sizeof(KeyT); // 16-byte loads
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    // 16-byte loads need a 16-byte aligned array (begin is a multiple of TILE); an unaligned array (a sub-range handed
    // in by the caller) takes the element-wise tail loop for everything
    const bool vec_ok = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint64_t nvec = vec_ok ? (end - begin) / VEC : 0;
    const VecT* vkeys = reinterpret_cast<const VecT*>(keys + begin);
    const KeyCodec<KeyT, XF> codec_in(xform & 3u);
    auto dig = [&](KeyT k) { return digit_of<KeyT>(codec_in.encode(k), shift, MASK); };
    auto tally_vec = [&](auto peel, const VecT& a) {
        if constexpr (sizeof(KeyT) == 4)
        {
            if (COLLECT) acc_or |= a.x | a.y | a.z | a.w, acc_and &= a.x & a.y & a.z & a.w;
            tally(peel, dig(a.x)); tally(peel, dig(a.y)); tally(peel, dig(a.z)); tally(peel, dig(a.w));
        }
        else
        {
            if (COLLECT) acc_or |= a.x | a.y, acc_and &= a.x & a.y;
            tally(peel, dig(a.x)); tally(peel, dig(a.y));
        }
    };
    uint64_t vbase = 0;
    // (the loop three times, chosen once per wave from its first keys: see wave_tally / wave_tally_runs)
    auto main_loop = [&](auto peel) {
        for (; vbase + 4 * THREADS <= nvec; vbase += 4 * THREADS) // block-uniform trip count, 4 x 16 B in flight per lane
        {
            if (wave_tally_gives_up(peel, run)) break; // (peel mode only; the wave comes back in the stateless mode)
            // non-temporal loads: the keys are read once; leaving them out of the Infinity Cache keeps the dirty lines of the
            // scatter that ran just before from being evicted under this kernel (0.255 -> 0.22 ms behind a scatter, 0.19 ->
            // 0.17 ms alone at 2^28 keys)
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
    if (4 * THREADS <= nvec) wave_tally_dispatch(dig(vkeys[tid].x), lane, run, main_loop, add_count);
    // tail (< 4 * THREADS vectors + a partial vector): plain per-key atomics, lanes may be inactive
    uint64_t i = begin + vbase * VEC + tid;
    for (; i + 7ull * THREADS < end; i += 8ull * THREADS) // 8 loads in flight per lane
    {
        KeyT k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = keys[i + (uint64_t) j * THREADS];
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            if (COLLECT) acc_or |= k[j], acc_and &= k[j];
            atomicAdd(&my_hist[dig(k[j])], 1u);
        }
    }
    for (; i < end; i += THREADS)
    {
        const KeyT k = keys[i];
        if (COLLECT) acc_or |= k, acc_and &= k;
        atomicAdd(&my_hist[dig(k)], 1u);
    }
    __syncthreads();
    if (COLLECT && (plan_flags & kPlanCollectBits)) plan_publish_bits<KeyT>(plan, acc_or, acc_and, lane);

    for (int d = tid; d < RADIX; d += THREADS)
    {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) c += hist[w][d];
        table[(size_t) d * gridDim.x + blockIdx.x] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: one workgroup per digit row: in-place exclusive scan of table[d][0..num_blocks), row total to totals[d].
// (The reference scans 16 partitions of nbp2 entries with 2*log2(nbp2) dispatches and keeps the 16 digit
//  totals in a separate buffer that the reorder shader scans itself: RadixSort.hpp:148-152, 311.)
// ---------------------------------------------------------------------------------------------------------
// pair_limit > 0 (the leader of a pair of passes): a table entry above it sends the follower back to its own count kernel.
// coarse != nullptr: every coarse_step-th scanned entry is also written to coarse[row][i / coarse_step].
template<int THREADS>
__global__ __launch_bounds__(THREADS) void radix_row_scan_kernel(uint32_t* __restrict__ table,
                                                                 uint32_t* __restrict__ totals, uint32_t num_blocks,
                                                                 uint32_t n = 0, PassPlan* plan = nullptr,
                                                                 uint32_t pass = 0, uint32_t pair_limit = 0,
                                                                 PassPlan* pair_plan = nullptr,
                                                                 uint32_t* __restrict__ coarse = nullptr, uint32_t coarse_step = 1)
{
    constexpr int WAVES = THREADS / kWave;
    __shared__ uint32_t wave_sums[WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the count kernel before this one found the pass an identity without counting (plan_digit_is_constant), or the pass belongs
    // to the sequence not taken: no table.  (`plan` is only given when the scan may MARK the pass an identity -- not for the encode /
    // decode passes of typed keys --; pair_plan is the plan whenever there is one: ADVICE r4)
    const PassPlan* any_plan = plan ? plan : pair_plan;
    if (any_plan && any_plan->skip[pass] == kSkipWithoutCounting) return;
    uint32_t* row = table + (size_t) blockIdx.x * num_blocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_blocks; base += THREADS)
    {
        uint32_t i = base + tid;
        uint32_t v = i < num_blocks ? row[i] : 0;
        if (pair_limit && v > pair_limit) pair_plan->pair_fallback[pass + 1] = 1;
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(v, lane, wtotal);
        if (lane == 0) wave_sums[wave] = wtotal;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++)
        {
            uint32_t s = wave_sums[w];
            if ((uint32_t) w < wave) woff += s;
            all += s;
        }
        if (i < num_blocks)
        {
            row[i] = carry + woff + excl;
            // (4-bit leader of a pair of passes: the table is per sub-block; the scatter's per-block table is every
            // coarse_step-th scanned entry)
            if (coarse && i % coarse_step == 0) coarse[(size_t) blockIdx.x * (num_blocks / coarse_step) + i / coarse_step] = carry + woff + excl;
        }
        carry += all;
        __syncthreads();
    }
    if (tid == 0)
    {
        totals[blockIdx.x] = carry;
        if (plan && carry == n) plan->skip[pass] = 1; // every key has this digit value: the pass permutes nothing
    }
}

// ---------------------------------------------------------------------------------------------------------
// (Measured alternative, not used by the library: see CHAINED in radix_scatter_kernel.)
// Mid-size sorts (one tile per workgroup, all workgroups resident): ONE read of the keys for the digit totals of every
// pass (k_radix_sort_counting_shader's global_count, RadixSort.hpp:52-56, for all passes at once), after which every pass
// is a single launch of the chained scatter kernel.  ghist[p][d] += this tile's keys with value d in digit p (global
// atomics: RADIX per pass and workgroup); the workgroup also clears its rows of the passes' chain words.
// shifts / widths: one byte per pass.
// ---------------------------------------------------------------------------------------------------------
template<int THREADS, int TILE, int MAX_PASSES = 4>
__global__ __launch_bounds__(THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ ghist,
                                                             uint32_t* __restrict__ chain, uint32_t n, uint32_t passes,
                                                             uint32_t shifts, uint32_t widths)
{
    constexpr int RADIX = 256;
    __shared__ uint32_t hist[MAX_PASSES][RADIX];
    const uint32_t tid = threadIdx.x, b = blockIdx.x, tiles = gridDim.x;
    for (int i = tid; i < MAX_PASSES * RADIX; i += THREADS) (&hist[0][0])[i] = 0;
    // per pass: local[d][row_words] (one word per tile) then group[d][grow_words] (one word per group of 16 tiles)
    const uint32_t row_words = (tiles + 15u) & ~15u, grow_words = (row_words / 16u + 15u) & ~15u;
    const size_t pass_words = (size_t) RADIX * (row_words + grow_words);
    for (uint32_t p = 0; p < passes; p++)
        for (uint32_t i = tid; i < (uint32_t) RADIX; i += THREADS)
        {
            chain[p * pass_words + (size_t) i * row_words + b] = 0u;
            if (b % 16u == 0u) chain[p * pass_words + (size_t) RADIX * row_words + (size_t) i * grow_words + b / 16u] = 0u;
        }
    __syncthreads();
    const uint64_t base = (uint64_t) b * TILE;
    for (uint32_t i = tid; i < (uint32_t) TILE; i += THREADS)
    {
        if (base + i < n)
        {
            const uint32_t k = keys[base + i];
            for (uint32_t p = 0; p < passes; p++)
                atomicAdd(&hist[p][__builtin_amdgcn_ubfe(k, (shifts >> (8 * p)) & 255u, (widths >> (8 * p)) & 255u)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < passes * RADIX; i += THREADS)
    {
        const uint32_t c = (&hist[0][0])[i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

constexpr uint32_t kFusedScanMaxBlocks = 32; // FUSED_SCAN scatter: every workgroup reads RADIX x nb counts itself, at most this many per thread
                                             // (THREADS / RADIX threads share a row; the host raises the limit for 4-bit digits: launch_pass)

// ---------------------------------------------------------------------------------------------------------
// K4: stable scatter of (key, val) by one digit.
// ---------------------------------------------------------------------------------------------------------

// LDS arrays of (key, val) pairs.  32-bit keys travel with their value as one 8-byte element (one ds_write_b64 /
// ds_read_b64 per pair); 64-bit keys use two arrays.
// VALS = false (keys-only sorts): keys alone, half (32-bit keys) or two thirds (64-bit keys) of the LDS per element,
// which the keys-only geometries spend on larger tiles.
template<typename KeyT, int COUNT, bool VALS = true>
struct PairArray
{
    KeyT keys[COUNT];
    __device__ __forceinline__ void put(uint32_t pos, KeyT k, uint32_t) { keys[pos] = k; }
    __device__ __forceinline__ void get(uint32_t pos, KeyT& k, uint32_t& v) const
    {
        k = keys[pos];
        v = 0u;
    }
};

template<int COUNT>
struct PairArray<uint32_t, COUNT, true>
{
    uint2 kv[COUNT];
    __device__ __forceinline__ void put(uint32_t pos, uint32_t k, uint32_t v) { kv[pos] = make_uint2(k, v); }
    __device__ __forceinline__ void get(uint32_t pos, uint32_t& k, uint32_t& v) const
    {
        const uint2 e = kv[pos];
        k = e.x;
        v = e.y;
    }
};

template<int COUNT>
struct PairArray<uint64_t, COUNT, true>
{
    uint64_t keys[COUNT];
    uint32_t vals[COUNT];
    __device__ __forceinline__ void put(uint32_t pos, uint64_t k, uint32_t v)
    {
        keys[pos] = k;
        vals[pos] = v;
    }
    __device__ __forceinline__ void get(uint32_t pos, uint64_t& k, uint32_t& v) const
    {
        k = keys[pos];
        v = vals[pos];
    }
};

#ifndef GLU_CARRY_ELEMS
#define GLU_CARRY_ELEMS 16 // tuning builds may override (tools/scatter_bench.hip)
#endif
constexpr int kBlockElems = GLU_CARRY_ELEMS; // 16 x 4 B = the 64-byte write block the carry keeps whole

// Row padding of the wave-private counter table: the offsets scan reads counters (wave 4q + k, digit) with q varying
// fastest across lanes; with rows of RADIX words all q land in the same bank (4-way / 2-way conflicts for 16 / 8 waves).
// Rows padded to RADIX + 16 / (WAVES / 4) words make those accesses conflict-free.
constexpr int wcnt_row_pad(int waves) { return waves >= 16 ? 4 : waves >= 8 ? 8 : 0; }

template<typename KeyT, int BITS, int THREADS, int KPT, bool CARRY, int ROUNDS = 1, bool VALS = true>
struct ScatterSmem
{
    static constexpr int RADIX = 1 << BITS;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    static constexpr int STAGE = TILE / ROUNDS;                  // ranked positions staged per round
    static_assert(KPT % ROUNDS == 0, "a round writes out KPT / ROUNDS positions per thread");
    PairArray<KeyT, STAGE, VALS> stage;                          // one round of the tile in ranked order
    PairArray<KeyT, CARRY ? RADIX * kBlockElems : 1, VALS> carry; // per digit: elements of a not yet complete 64-B block
    static constexpr int WCNT_STRIDE = RADIX + wcnt_row_pad(WAVES);
    uint32_t wcnt[WAVES][WCNT_STRIDE]; // wave-private running digit counters -> first ranked position of (wave, digit)
    uint32_t tstart[RADIX];      // first ranked position of each digit in the tile
    uint2 dest[RADIX];           // .x: global index = ranked position + dest[digit].x;  .y (CARRY): elements with a
                                 // global index >= dest[digit].y go to the carry, not to memory (one ds_read_b64 for both)
    uint2 flush[RADIX];          // CARRY: carried elements [.x, .y) of the digit are written this tile
    uint32_t scan_tmp[WAVES];
};

// Per tile: load keys + values (wave-striped) -> rank inside each wave (ballot match + wave-private LDS counters)
// -> one block-wide exclusive scan of the [digit][wave] counters -> stage (key, val) in LDS in ranked order ->
// write out with consecutive threads on consecutive ranked positions.
//
// CARRY: a digit's run of one tile rarely ends on a 64-byte boundary, and a partially written 64-byte block that is
// completed a whole tile period later costs the memory system a fill read + a second write (measured: +50 % fetch
// traffic with 256 digits).  With CARRY the tail of every run (the elements past the last 64-byte boundary) stays
// in LDS and is written, together with the head of the digit's next run, when its block is complete -- the only
// partial blocks left are at the two ends of a workgroup's range.
//
// Template parameters beyond the geometry (the library instantiates the defaults except XF / VALS / FUSED_SCAN):
//   XF          the pass encodes keys on load / decodes them on store (first / last pass of a signed or float sort);
//   VALS        false: keys-only sort, no value arrays, LDS arrays of keys alone;
//   FUSED_SCAN  the prologue sums the raw count table itself (<= kFusedScanMaxBlocks workgroups, no row-scan launch);
//   ABLATE      tuning builds: 1 = write every tile back linearly (prices the scattered stores; wrong results),
//               2 = only a workgroup's first tile is loaded (prices the exposed load phase), 4 = no carry flush and no
//               write-out (prices them), 5 = both (what rank + offsets + staging cost alone); all wrong results,
//               2 = only the first tile of a workgroup is loaded (prices the exposed load phase; wrong results),
//               3 = stagger the workgroups' start;  STAMPS: wave 0 adds the s_memtime cycles of each phase to stamps[0..7];
//   ROUNDS > 1  tile = ROUNDS x the staging area (keys, values, ranks stay in registers across the staging rounds);
//   PREFETCH    the next tile's loads are issued one pair per rank iteration;  DMA: tile loads as LDS-DMA into the idle
//               staging area.  ROUNDS / PREFETCH / DMA are measured alternatives kept for tools/scatter_bench.hip
//               (level with or slower than the defaults), not used by the library.
template<typename KeyT, int BITS, int THREADS, int KPT, bool CARRY = true, int ABLATE = 0, bool STAMPS = false,
         int MIN_WAVES_PER_SIMD = 1, int ROUNDS = 1, bool PREFETCH = false, bool DMA = false, bool XF = false, bool VALS = true, bool FUSED_SCAN = false,
         bool CHAINED = false>
__global__ __launch_bounds__(THREADS, MIN_WAVES_PER_SIMD) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_a, const uint32_t* __restrict__ vals_a, KeyT* __restrict__ keys_b,
    uint32_t* __restrict__ vals_b, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
    uint32_t n, uint32_t shift, uint32_t mask, uint32_t tiles_total, unsigned long long* stamps = nullptr,
    uint32_t xform = 0, PassPlan* plan = nullptr, uint32_t pass = 0, uint32_t* __restrict__ chain = nullptr)
{
    // CHAINED -- a measured alternative kept for tools/chained_probe.hip, NOT used by the library (one
    // cross-CU hand-off costs 2.5-3.6 us on this chip, more than the count and row-scan launches it replaces; a sort of 2^20
    // pairs took 186 us this way against 67 us).  Mid-size sorts, one tile per workgroup, every workgroup resident at once: no count kernel and no row scan.
    // `totals` are the digit totals of this pass from the up-front histogram kernel (radix_hist_kernel); a workgroup's
    // offset inside every digit is the sum of the digit counts of the tiles before it, which every workgroup publishes
    // (chain[tile * RADIX + d] = count + 1; 0 = not yet) as soon as it has ranked its tile.
    static_assert(!CHAINED || (!CARRY && !FUSED_SCAN && ROUNDS == 1 && !PREFETCH), "chained passes run the plain one-tile form");
    // Source and destination arrays: (a -> b) as passed, or (b -> a) when the plan says that the data sits in b.  The
    // body only ever touches memory through the four local pointers below.
    const KeyT* __restrict__ src_keys = keys_a;
    const uint32_t* __restrict__ src_vals = vals_a;
    KeyT* __restrict__ dst_keys = keys_b;
    uint32_t* __restrict__ dst_vals = vals_b;
    if (plan)
    {
        const uint32_t flip = pass > 0 ? plan->flip[pass] : 0u;
        const uint32_t skip = plan->skip[pass];
        if (blockIdx.x == 0 && threadIdx.x == 0) plan->flip[pass + 1] = flip ^ (skip ? 0u : 1u);
        if (skip) return; // identity pass (kernel-uniform)
        if (flip)
        {
            src_keys = keys_b;
            src_vals = vals_b;
            dst_keys = const_cast<KeyT*>(keys_a);
            dst_vals = const_cast<uint32_t*>(vals_a);
        }
    }
    using Smem = ScatterSmem<KeyT, BITS, THREADS, KPT, CARRY, ROUNDS, VALS>;
    constexpr int STAGE = Smem::STAGE;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int TILE = Smem::TILE;
    constexpr int WAVE_TILE = kWave * KPT;
    constexpr int WQ = WAVES / 4;              // threads per digit in the offset scan (4 waves each)
    constexpr int SCAN_THREADS = RADIX * WQ;   // threads taking part in the offset scan
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    constexpr uint32_t BLK = kBlockElems;
    const uint32_t MASK = mask; // <= RADIX - 1
    static_assert(WAVES % 4 == 0, "offset scan handles 4 waves per thread");
    static_assert(SCAN_THREADS <= THREADS && RADIX <= THREADS, "offset phase needs RADIX * WAVES / 4 threads");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    constexpr bool has_vals = VALS; // keys-only sorts run the VALS = false instantiation (no value arrays, keys-only LDS)
    const bool dma_ok = ((reinterpret_cast<uintptr_t>(src_keys) | reinterpret_cast<uintptr_t>(src_vals)) & 15u) == 0;
    const KeyCodec<KeyT, XF> codec_in(xform & 3u), codec_out((xform >> 2) & 3u); // key encode on load / decode on store

    // ---- prologue: this workgroup's global base for every digit:
    //      exclusive scan of the digit totals (RadixSort.hpp:148-152) + this block's scanned table entry (:176)
    //      FUSED_SCAN (few workgroups: launch-bound sizes): `table` holds the raw counts of the count kernel and every
    //      workgroup sums its own row prefixes and the row totals, which saves the row-scan launch of the pass.
    uint32_t digit_base = 0; // valid in threads tid < RADIX: global index of the digit's next element
    {
        uint32_t t = 0, before = 0;
        if (FUSED_SCAN)
        {
            // THREADS / RADIX threads share a digit's row: thread (part, d) sums the entries [part * chunk, + chunk) -- at most
            // kFusedScanMaxBlocks independent loads per thread (measured against 16 guarded loads in flight and against a
            // block-major table: this plain loop wins; twice as many loads per thread cost more than the row-scan launch
            // they save) -- and thread d adds the parts up.  With 4-bit digits 32 threads share a row, and sorts of up to 256
            // workgroups run without the row-scan kernel (the host decides: launch_pass).
            constexpr uint32_t PARTS = THREADS / RADIX >= 1 ? THREADS / RADIX : 1;
            uint32_t* scratch = reinterpret_cast<uint32_t*>(&s.stage); // (the staging area is not in use yet)
            static_assert(sizeof(s.stage) >= 2 * THREADS * sizeof(uint32_t), "prologue scratch");
            const uint32_t part = tid / RADIX, d = tid % RADIX;
            if (part < PARTS)
            {
                const uint32_t chunk = (nb + PARTS - 1) / PARTS;
                const uint32_t j0 = part * chunk, j1 = j0 + chunk < nb ? j0 + chunk : nb;
                const uint32_t* row = table + (size_t) d * nb;
                uint32_t tp = 0, bp = 0;
                for (uint32_t j = j0; j < j1; j++)
                {
                    const uint32_t c = row[j];
                    tp += c;
                    bp += j < b ? c : 0u;
                }
                scratch[tid] = tp;
                scratch[THREADS + tid] = bp;
            }
            __syncthreads();
            if (tid < RADIX)
            {
#pragma unroll
                for (uint32_t k = 0; k < PARTS; k++)
                {
                    t += scratch[k * RADIX + tid];
                    before += scratch[THREADS + k * RADIX + tid];
                }
            }
            __syncthreads(); // the staging area is free again
        }
        else
            t = tid < RADIX ? totals[tid] : 0;
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(t, lane, wtotal);
        if (lane == 0) s.scan_tmp[wave] = wtotal;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++)
            if ((uint32_t) w < wave) woff += s.scan_tmp[w];
        if (tid < RADIX) digit_base = woff + excl + (CHAINED ? 0u : FUSED_SCAN ? before : table[(size_t) tid * nb + b]);
    }
    uint32_t carry_start = digit_base; // CARRY: elements [carry_start, digit_base) of the digit are held in s.carry
    for (int i = tid; i < WAVES * Smem::WCNT_STRIDE; i += THREADS) (&s.wcnt[0][0])[i] = 0;
    __syncthreads();

    uint32_t first, last;
    block_tile_range(b, nb, tiles_total, first, last);
    if (ABLATE == 3)
    {
        // experiment: spread the workgroups' phases over one tile time so that the chip is not in lockstep
        for (uint32_t k = 0; k < (b * 7u) % 16u; k++) __builtin_amdgcn_s_sleep(30);
        __syncthreads();
    }

    // wave-striped layout: item i of lane l of wave w is element w*WAVE_TILE + i*64 + l of the tile, so that
    // "item-major, then lane" order inside a wave is memory order -> ranks are stable
    const uint32_t wave_off = wave * WAVE_TILE + lane;

    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int slot) {
        if (STAMPS)
        {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            acc[slot] += t - tprev;
            tprev = t;
        }
    };

    // writes the carried elements [flush_lo, flush_hi) of every digit: thread (digit, slot) pairs, 16 consecutive
    // lanes per digit
    auto flush_carry = [&]() {
        // batched: all LDS reads of the FJ (digit, slot) pairs of this thread are issued before any dependent work
        constexpr int FJ = (RADIX * (int) BLK + THREADS - 1) / THREADS;
        uint32_t lo[FJ], hi[FJ];
        KeyT fk[FJ];
        uint32_t fv[FJ];
#pragma unroll
        for (int j = 0; j < FJ; j++)
        {
            uint32_t e = j * THREADS + tid;
            if (RADIX * BLK % THREADS != 0 && e >= RADIX * BLK) e = RADIX * BLK - 1; // clamp (result unused)
            const uint2 f = s.flush[e / BLK];
            lo[j] = f.x;
            hi[j] = f.y;
            s.carry.get(e, fk[j], fv[j]);
        }
#pragma unroll
        for (int j = 0; j < FJ; j++)
        {
            const uint32_t e = j * THREADS + tid;
            const uint32_t g = (lo[j] & ~(BLK - 1)) + (e % BLK);
            const bool live = g >= lo[j] && g < hi[j] && (RADIX * BLK % THREADS == 0 || e < RADIX * BLK) &&
                              (ABLATE == 0 || g < n); // ablation builds scatter with made-up keys: stay inside the arrays
            if (live)
            {
                dst_keys[g] = codec_out.decode(fk[j]);
                if (has_vals) dst_vals[g] = fv[j];
            }
        }
    };

    // loads one tile into registers (wave-striped); positions past the end of the array read as pads
    KeyT key[KPT];
    uint32_t val[KPT];
    auto load_tile = [&](uint32_t t) {
        const uint64_t base = (uint64_t) t * TILE;
        const uint64_t left = (uint64_t) n - base;
        if (DMA && VALS && dma_ok && left >= (uint64_t) TILE)
        {
            // Full tile of 16-byte aligned arrays: every wave copies ITS 64 * KPT keys and values into the (idle) staging
            // area with 1 KiB LDS-DMA pieces (global_load_lds_dwordx4: no VGPRs, 4x fewer vector-memory instructions
            // than dword loads, which are issue-bound here), then reads them back wave-striped.  The raw image is
            // wave-private, so no workgroup barrier is needed between the copy and the reads.
            unsigned char* raw = reinterpret_cast<unsigned char*>(&s.stage);
            const KeyT* rawk = reinterpret_cast<const KeyT*>(raw) + wave * WAVE_TILE;
            const uint32_t* rawv = reinterpret_cast<const uint32_t*>(raw + (size_t) TILE * sizeof(KeyT)) + wave * WAVE_TILE;
            const unsigned char* gk = reinterpret_cast<const unsigned char*>(src_keys + base + wave * WAVE_TILE);
            const unsigned char* gv = reinterpret_cast<const unsigned char*>(src_vals + base + wave * WAVE_TILE);
            constexpr int KEY_PIECES = KPT * (int) sizeof(KeyT) / 16, VAL_PIECES = KPT * 4 / 16;
            static_assert(!DMA || ((KPT * sizeof(KeyT)) % 16 == 0 && (KPT * 4) % 16 == 0), "whole 1 KiB pieces per wave");
#pragma unroll
            for (int j = 0; j < KEY_PIECES; j++)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*) (gk + j * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*) (reinterpret_cast<const unsigned char*>(rawk) + j * 1024), 16, 0, GLU_DMA_AUX);
            if (has_vals)
            {
#pragma unroll
                for (int j = 0; j < VAL_PIECES; j++)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*) (gv + j * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void*) (reinterpret_cast<const unsigned char*>(rawv) + j * 1024), 16, 0, GLU_DMA_AUX);
            }
            if (STAMPS) stamp(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < KPT; i++) key[i] = codec_in.encode(rawk[i * kWave + lane]);
#pragma unroll
            for (int i = 0; i < KPT; i++) val[i] = has_vals ? rawv[i * kWave + lane] : 0u;
            if (STAMPS)
            {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp(1);
            }
        }
        else if (left >= (uint64_t) TILE)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++) key[i] = codec_in.encode(src_keys[base + wave_off + i * kWave]);
#pragma unroll
            for (int i = 0; i < KPT; i++) val[i] = has_vals ? src_vals[base + wave_off + i * kWave] : 0u;
        }
        else
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t p = wave_off + i * kWave;
                const bool ok = p < (uint32_t) left;
                // pad: last digit, after all real keys
                key[i] = ok ? codec_in.encode(src_keys[base + p]) : (KeyT) ~(KeyT) 0;
                val[i] = (ok && has_vals) ? src_vals[base + p] : 0u;
            }
        }
    };
    // PREFETCH (experimental, off): the first tile is loaded here, every later one while its predecessor is ranked
    if (PREFETCH && first < last) load_tile(first);

    for (uint32_t tile = first; tile < last; tile++)
    {
        const uint64_t tile_base = (uint64_t) tile * TILE;
        const uint64_t rem = (uint64_t) n - tile_base;
        const uint32_t tile_valid = rem < (uint64_t) TILE ? (uint32_t) rem : (uint32_t) TILE;
        if (STAMPS) tprev = __builtin_amdgcn_s_memtime();

        // ---- load (PREFETCH: already issued while the previous tile was being ranked)
        if (!PREFETCH && ((ABLATE != 2 && ABLATE != 5) || tile == first)) load_tile(tile); // ABLATE 2 / 5: tile loads are free
        if (STAMPS)
        {
            stamp(0); // issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(1); // load latency (diagnostic builds wait here; production waits at first use)
        }

        // PREFETCH: the next tile's loads are issued one pair per rank iteration (into their own registers), so that
        // they neither wait for a full vector-memory queue nor sit between this tile's stores
        const bool sprinkle = PREFETCH && tile + 1 < last && (uint64_t) n - (tile_base + TILE) >= (uint64_t) TILE;
        KeyT nk[PREFETCH ? KPT : 1];
        uint32_t nv[PREFETCH ? KPT : 1];

        // ---- rank inside the wave
        uint32_t rank[KPT];
        uint32_t* my_cnt = s.wcnt[wave];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if (PREFETCH && sprinkle)
            {
                const uint64_t nbase = tile_base + TILE + wave_off;
                nk[i] = src_keys[nbase + i * kWave];
                nv[i] = has_vals ? src_vals[nbase + i * kWave] : 0u;
            }
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            uint32_t* const cnt = my_cnt + d;
            const uint32_t prev = *cnt; // issued first: its LDS latency hides under the ballots below
            // peers = lanes whose digit equals mine.  Per digit bit: sel = 0 / ~0 (v_bfe_i32), m = ballot(bit set),
            // peers &= ~(m ^ sel) -- one v_bitop3_b32 per 32-bit half (truth table 0x90: a & ~(b ^ c)).
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < BITS; bit++)
            {
                int32_t sel; // asm: keeps bit 0 from being rewritten as -(d & 1) and a compare chain (6 instructions)
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = (uint32_t) __popc(plo) + (uint32_t) __popc(phi);
            rank[i] = prev + lower;
            asm volatile("" : "+v"(rank[i])); // keep the rank (1 register), not the two peer masks, live across the phases
            // every peer stores the same new count to the same address (no exec-mask juggling for a leader lane)
            *cnt = prev + total;
        }
        stamp(2); // rank
        __syncthreads();
        stamp(3); // barrier after rank

        // ---- offsets: exclusive scan of the counters in (digit, wave) order = first ranked position of every
        //      (wave, digit) group.  Thread t < RADIX * WAVES/4 owns digit t / (WAVES/4), waves 4*(t % (WAVES/4)) .. +3.
        {
            const uint32_t sd = tid / WQ, sw = (tid % WQ) * 4;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (tid < SCAN_THREADS)
            {
                c0 = s.wcnt[sw + 0][sd];
                c1 = s.wcnt[sw + 1][sd];
                c2 = s.wcnt[sw + 2][sd];
                c3 = s.wcnt[sw + 3][sd];
            }
            uint32_t excl = 0;
            if (wave < SCAN_WAVES) // wave-uniform
            {
                uint32_t wtotal;
                excl = wave_exclusive_sum(c0 + c1 + c2 + c3, lane, wtotal);
                if (SCAN_WAVES > 1 && lane == 0) s.scan_tmp[wave] = wtotal;
            }
            if (SCAN_WAVES > 1)
            {
                __syncthreads();
                excl += sum_of_preceding_waves(s.scan_tmp, SCAN_WAVES, wave, lane);
            }
            if (tid < SCAN_THREADS)
            {
                s.wcnt[sw + 0][sd] = excl;
                s.wcnt[sw + 1][sd] = excl + c0;
                s.wcnt[sw + 2][sd] = excl + c0 + c1;
                s.wcnt[sw + 3][sd] = excl + c0 + c1 + c2;
                if (sw == 0) s.tstart[sd] = excl;
            }
        }
        __syncthreads();
        stamp(4); // offsets

        if (CHAINED)
        {
            // Publish this tile's digit counts, then add up those of every tile before it -- in two levels, so that a tile reads
            // at most 15 + tiles / 16 words per digit instead of one per predecessor (reading them all cost 33 MB of polling loads
            // per pass at 256 tiles: 60 us per pass): tiles form groups of 16; the last tile of a group publishes the group's
            // total; a tile adds the totals of the groups before its own and the counts of the tiles before it inside its
            // group.  chain = local[d][row_words] then group[d][grow_words]; a word is value + 1, 0 = not yet.  All
            // workgroups are resident and publish before they wait, and a group total waits for plain counts only: the waits end.
            constexpr uint32_t PARTS = THREADS / RADIX >= 1 ? THREADS / RADIX : 1;
            constexpr uint32_t GROUP = 16;
            const uint32_t row_words = (tiles_total + GROUP - 1) & ~(GROUP - 1);
            const uint32_t grow_words = (row_words / GROUP + GROUP - 1) & ~(GROUP - 1);
            uint32_t* const local = chain;
            uint32_t* const group = chain + (size_t) RADIX * row_words;
            uint32_t* scratch = reinterpret_cast<uint32_t*>(&s.stage); // (the staging area is not in use yet)
            uint32_t len = 0;
            if (tid < RADIX)
            {
                const uint32_t ts = s.tstart[tid];
                const uint32_t te = tid + 1 < RADIX ? s.tstart[tid + 1] : (uint32_t) TILE;
                len = te - ts;
                if (tid == MASK) len -= (uint32_t) TILE - tile_valid;
                __hip_atomic_store(&local[(size_t) tid * row_words + b], len + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // 16 independent loads of one aligned group of words, repeated until the first `want` of them are published
            auto sum_ready = [&](const uint32_t* words, uint32_t want) {
                uint32_t spins = 0;
                for (;;)
                {
                    uint32_t v[GROUP];
#pragma unroll
                    for (uint32_t j = 0; j < GROUP; j++)
                    {
#ifdef GLU_CHAIN_POLL_RMW
                        v[j] = __hip_atomic_fetch_add(const_cast<uint32_t*>(&words[j]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                        v[j] = __hip_atomic_load(&words[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                    }
                    bool ready = true;
                    uint32_t total = 0;
#pragma unroll
                    for (uint32_t j = 0; j < GROUP; j++)
                    {
                        ready = ready && (j >= want || v[j] != 0u);
                        total += j < want ? v[j] - 1u : 0u;
                    }
                    if (ready) return total;
                    if (++spins > (1u << 22)) __builtin_trap(); // fail loudly, never hang
                    __builtin_amdgcn_s_sleep(1);
                }
            };
            const uint32_t part = tid / RADIX, d = tid % RADIX;
            const uint32_t my_group = b / GROUP;
            uint32_t sum = 0;
            if (part == 0)
            {
                // the tiles before this one inside its group; the group's last tile owes the group's total
                if (b % GROUP) sum = sum_ready(local + (size_t) d * row_words + my_group * GROUP, b % GROUP);
                if (b % GROUP == GROUP - 1)
                    __hip_atomic_store(&group[(size_t) d * grow_words + my_group], sum + len + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the groups before this one (shared by the other threads of the digit; by the same thread when there is only one)
            if (part < PARTS && (part > 0 || PARTS == 1))
                for (uint32_t g0 = (PARTS == 1 ? 0u : part - 1u) * GROUP; g0 < my_group; g0 += (PARTS == 1 ? 1u : PARTS - 1u) * GROUP)
                    sum += sum_ready(group + (size_t) d * grow_words + g0, my_group - g0 < GROUP ? my_group - g0 : GROUP);
            if (part < PARTS) scratch[tid] = sum;
            __syncthreads();
            if (tid < RADIX)
            {
#pragma unroll
                for (uint32_t k = 0; k < PARTS; k++) digit_base += scratch[k * RADIX + tid];
            }
            __syncthreads(); // the staging area is free again
        }
        // ---- thread d: where digit d of this tile goes, advance the running base
        if (tid < RADIX)
        {
            const uint32_t ts = s.tstart[tid];
            const uint32_t te = tid + 1 < RADIX ? s.tstart[tid + 1] : (uint32_t) TILE;
            uint32_t len = te - ts;
            // pads (last, partial tile only) were ranked at the end of the highest used digit: not real elements
            if (tid == MASK) len -= (uint32_t) TILE - tile_valid;
            uint2 dst;
            dst.x = digit_base - ts;
            dst.y = 0xFFFFFFFFu;
            if (CARRY)
            {
                const uint32_t end = digit_base + len;
                const uint32_t aligned_end = end & ~(BLK - 1);
                if (aligned_end > carry_start)
                {
                    // the run reaches past a 64-byte boundary: the carried elements are completed -> write them,
                    // write the run up to its last boundary, carry its tail
                    s.flush[tid] = make_uint2(carry_start, digit_base);
                    dst.y = aligned_end;
                    carry_start = aligned_end;
                }
                else
                {
                    // still inside the same 64-byte block: everything joins the carry
                    s.flush[tid] = make_uint2(0u, 0u);
                    dst.y = digit_base;
                }
            }
            s.dest[tid] = dst;
            digit_base += len;
        }
        // ---- ranked position of every item (rank[] becomes the position inside the tile)
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            rank[i] += my_cnt[digit_of<KeyT>(key[i], shift, MASK)];
        }

#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
            // ---- stage the ranked positions [r * STAGE, (r + 1) * STAGE)
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t pos = rank[i] - (uint32_t) (r * STAGE);
                if (ROUNDS == 1 || pos < (uint32_t) STAGE) s.stage.put(pos, key[i], val[i]);
                }
            __syncthreads();
            if (r == 0) stamp(5); // stage + barrier
            if (CARRY && r == 0)
            {
                if (ABLATE < 4) flush_carry(); // old carry out before the write-out below refills the slots
                __syncthreads();
            }

            // ---- write out: consecutive threads -> consecutive ranked positions -> (per digit) consecutive addresses.
            //      Batched sweeps (staged pairs, per-digit lookups, stores) so that the LDS latencies overlap instead of
            //      being paid once per item behind a branch.
            constexpr int WI = KPT / ROUNDS;
#ifndef GLU_WRITE_BATCH
#define GLU_WRITE_BATCH 4 // items per write-out batch (tuning builds override; 0 = the whole round)
#endif
            // 4 items per batch: enough LDS reads in flight to cover their latency; the whole round in one batch costs
            // 5 registers per item and was 2 % (32-bit keys, 12 items) to 17 % (64-bit keys, 16 items) slower
            constexpr int WB = (GLU_WRITE_BATCH > 0 && WI % GLU_WRITE_BATCH == 0) ? GLU_WRITE_BATCH : WI;
#pragma unroll
            for (int b0 = 0; b0 < (ABLATE >= 4 ? 0 : WI); b0 += WB) // ABLATE 4 / 5: no write-out at all (prices it)
            {
                KeyT wk[WB];
                uint32_t wv[WB], wg[WB], wlim[WB];
#pragma unroll
                for (int i = 0; i < WB; i++) s.stage.get((b0 + i) * THREADS + tid, wk[i], wv[i]);
#pragma unroll
                for (int i = 0; i < WB; i++)
                {
                    const uint32_t wd = digit_of<KeyT>(wk[i], shift, MASK);
                    const uint2 dst = s.dest[wd];
                    wg[i] = r * STAGE + (b0 + i) * THREADS + tid + dst.x;
                    wlim[i] = dst.y;
                }
#pragma unroll
                for (int i = 0; i < WB; i++)
                {
                    const uint32_t p = r * STAGE + (b0 + i) * THREADS + tid; // ranked position inside the tile
                    const bool valid = p < tile_valid;
                    const bool to_carry = CARRY && wg[i] >= wlim[i];
                    uint32_t g = wg[i];
                    if (ABLATE == 1) g = (uint32_t) tile_base + p;
                    if (valid && !to_carry && (ABLATE == 0 || g < n))
                    {
                        dst_keys[g] = codec_out.decode(wk[i]);
                        if (has_vals) dst_vals[g] = wv[i];
                    }
                    if (CARRY && valid && to_carry)
                        s.carry.put(digit_of<KeyT>(wk[i], shift, MASK) * BLK + (wg[i] & (BLK - 1)), wk[i], wv[i]);
                }
            }
            // a partial next tile (the last of the array) takes the guarded loads
            if (PREFETCH && r == ROUNDS - 1 && tile + 1 < last)
            {
                if (sprinkle)
                {
#pragma unroll
                    for (int i = 0; i < KPT; i++)
                    {
                        key[i] = codec_in.encode(nk[i]);
                        val[i] = nv[i];
                    }
                }
                else
                    load_tile(tile + 1);
            }
            if (r + 1 < ROUNDS) __syncthreads(); // the next round overwrites the staging area
        }
        for (int i = tid; i < WAVES * Smem::WCNT_STRIDE; i += THREADS) (&s.wcnt[0][0])[i] = 0;
        stamp(6); // write-out issue
        __syncthreads();
        stamp(7); // final barrier
    }

    if (CARRY)
    {
        // what is still carried at the end of this workgroup's range: the (partial) last block of every digit
        if (tid < RADIX)
        {
            s.flush[tid] = make_uint2(carry_start, digit_base);
        }
        __syncthreads();
        flush_carry();
    }
    if (STAMPS && tid == 0 && stamps)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&stamps[i], acc[i]);
    }
}

// Planned sorts: if the data ended up in the scratch arrays (odd number of executed passes), bring it home.
template<typename KeyT, bool VALS>
__global__ __launch_bounds__(256) void radix_finalize_kernel(KeyT* __restrict__ keys_a, uint32_t* __restrict__ vals_a,
                                                             const KeyT* __restrict__ keys_b,
                                                             const uint32_t* __restrict__ vals_b, uint32_t n,
                                                             const PassPlan* plan, uint32_t passes, uint32_t attempt_passes = 0)
{
    // (a sort that ended in LDS: where its `attempt_passes` top-bit passes left the data -- the in-LDS pass works in place, and
    // the ordinary passes behind them, which return at once on a stream of their own, have not kept flip[] in order)
    if (!plan->flip[attempt_passes && plan->finish ? attempt_passes : passes]) return;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        keys_a[i] = keys_b[i];
        if (VALS) vals_a[i] = vals_b[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Whole sort in ONE workgroup for inputs that fit one tile (n <= THREADS * KPT): every pass ranks, scans and
// re-stages the tile in LDS; only the first load and the last store touch memory.  Replaces 3 launches per pass (the
// reference needs 32 dispatches for N = 1024, README: 0.66 ms) with a single launch.
// ---------------------------------------------------------------------------------------------------------
template<typename KeyT, int BITS, int THREADS, int KPT>
struct SingleBlockSmem
{
    static constexpr int RADIX = 1 << BITS;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    PairArray<KeyT, TILE> stage;
    static constexpr int WCNT_STRIDE = RADIX + wcnt_row_pad(WAVES);
    uint32_t wcnt[WAVES][WCNT_STRIDE];
    uint32_t scan_tmp[WAVES];
};

template<typename KeyT, int BITS, int THREADS, int KPT, bool XF = false>
__global__ __launch_bounds__(THREADS) void radix_sort_single_block_kernel(KeyT* __restrict__ keys,
                                                                          uint32_t* __restrict__ vals, uint32_t n,
                                                                          uint32_t first_bit, uint32_t end_bit,
                                                                          uint32_t xform = 0)
{
    using Smem = SingleBlockSmem<KeyT, BITS, THREADS, KPT>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int WAVE_TILE = kWave * KPT;
    constexpr int WQ = WAVES / 4;
    constexpr int SCAN_THREADS = RADIX * WQ;
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    static_assert(WAVES % 4 == 0 && SCAN_THREADS <= THREADS, "offset scan geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wave_off = wave * WAVE_TILE + lane;

    const KeyCodec<KeyT, XF> codec_in(xform & 3u), codec_out((xform >> 2) & 3u);
    KeyT key[KPT];
    uint32_t val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        const bool ok = p < n;
        key[i] = ok ? codec_in.encode(keys[p]) : (KeyT) ~(KeyT) 0; // pads: highest digit in every pass
        val[i] = (ok && vals) ? vals[p] : 0u;
    }

    uint32_t* my_cnt = s.wcnt[wave];
    // stable passes over the key bits [first_bit, end_bit), BITS at a time (a digit never straddles the two words of a
    // 64-bit key: digit_of picks one word)
    uint32_t bits = 0;
    for (uint32_t shift = first_bit; shift < end_bit; shift += bits)
    {
        bits = end_bit - shift < (uint32_t) BITS ? end_bit - shift : (uint32_t) BITS;
        if (sizeof(KeyT) == 8 && shift < 32u && shift + bits > 32u) bits = 32u - shift;
        const uint32_t MASK = (1u << bits) - 1;
        for (int i = tid; i < WAVES * Smem::WCNT_STRIDE; i += THREADS) (&s.wcnt[0][0])[i] = 0;
        __syncthreads();

        uint32_t rank[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            uint32_t* const cnt = my_cnt + d;
            const uint32_t prev = *cnt;
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < BITS; bit++)
            {
                int32_t sel; // asm: keeps bit 0 from being rewritten as -(d & 1) and a compare chain (6 instructions)
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = (uint32_t) __popc(plo) + (uint32_t) __popc(phi);
            rank[i] = prev + lower;
            asm volatile("" : "+v"(rank[i])); // keep the rank (1 register), not the two peer masks, live across the phases
            *cnt = prev + total;
        }
        __syncthreads();

        {
            const uint32_t sd = tid / WQ, sw = (tid % WQ) * 4;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (tid < SCAN_THREADS)
            {
                c0 = s.wcnt[sw + 0][sd];
                c1 = s.wcnt[sw + 1][sd];
                c2 = s.wcnt[sw + 2][sd];
                c3 = s.wcnt[sw + 3][sd];
            }
            uint32_t excl = 0;
            if (wave < SCAN_WAVES)
            {
                uint32_t wtotal;
                excl = wave_exclusive_sum(c0 + c1 + c2 + c3, lane, wtotal);
                if (SCAN_WAVES > 1 && lane == 0) s.scan_tmp[wave] = wtotal;
            }
            if (SCAN_WAVES > 1)
            {
                __syncthreads();
                excl += sum_of_preceding_waves(s.scan_tmp, SCAN_WAVES, wave, lane);
            }
            if (tid < SCAN_THREADS)
            {
                s.wcnt[sw + 0][sd] = excl;
                s.wcnt[sw + 1][sd] = excl + c0;
                s.wcnt[sw + 2][sd] = excl + c0 + c1;
                s.wcnt[sw + 3][sd] = excl + c0 + c1 + c2;
            }
        }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < KPT; i++)
            s.stage.put(my_cnt[digit_of<KeyT>(key[i], shift, MASK)] + rank[i], key[i], val[i]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; i++) s.stage.get(wave_off + i * kWave, key[i], val[i]);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        if (p < n)
        {
            keys[p] = codec_out.decode(key[i]);
            if (vals) vals[p] = val[i];
        }
    }
}

} // namespace glu_hip
