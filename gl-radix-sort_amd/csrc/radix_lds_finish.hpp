// radix_lds_finish.hpp -- a large whole-key sort that ENDS IN LDS: two counting passes and one in-LDS pass instead of four
// (32-bit keys) or eight (64-bit keys) counting passes.
//
// The reference sorts least significant digit first, every pass a full permutation of the array in memory
// (k_radix_sort_counting_shader + BlellochScan + k_radix_sort_reordering_shader per 4-bit step, glu/RadixSort.hpp:289-345);
// this library's large sort does the same with 8-bit passes: 4 x (16 B of scatter + its share of a key read) = 72.5 B per
// pair of 32-bit key and value.  The result of a stable sort does not depend on how it is reached, and the memory system
// prices bytes, not passes:
//
//   1. the two counting passes on 16 TOP key bits first -- digit [top - 16, top - 8), then [top - 8, top), an ordinary pair
//      of passes (radix_pair_passes.hpp); `top` is the key's width, or the highest key bit that varied in the object's last
//      attempt (a guess, checked below).  After them the array is a sequence of 65536 RUNS, run r = the pairs whose key has
//      the value r in those bits, in input order.  The leader's two-digit histogram T2 already holds every run's length:
//      len[r] = sum over blocks b of T2[r & 255][b][r >> 8].
//   2. if no run is longer than what one workgroup sorts in LDS (tiles of 1536 .. 9216 pairs: the device picks the smallest
//      of the enqueued geometries that holds the longest run) and no key bit from `top` up varies, ONE pass finishes the
//      sort in place: a workgroup per run orders it by the low top - 16 bits (rounds of 8 bits of rank / scan / re-stage, as
//      in radix_sort_single_block_kernel: two for 32-bit keys, six for 64-bit keys) and writes it back where it was.  16 B
//      per pair instead of the 2 x 20.5 of two more passes: 52.25 B per pair in all (64-bit keys: 80.25 instead of 225).
//      Typed keys (signed, float) are encoded on load by the first top-bit pass and decoded on store by this one.
//   3. otherwise (keys that crowd into few runs: small value ranges under the first guess, heavy duplicates) the top-bit
//      passes are not run at all and the passes of the ordinary sort follow.  The decision is made on the device, after the
//      leader's count kernel and before its scatter, from exact run lengths; the launch sequence is the same either way and
//      the kernels of the path not taken return at once (PassPlan::off, PassPlan::skip).  A refused attempt costs one read
//      of the keys (the leader's count kernel: 4 of 72.5 B per pair); its outcome and the key bits that varied reach the host
//      through a pinned word that the next sort call reads without synchronising (glu_radix_sort_s::finish_hint).
//
//   radix_finish_lengths_kernel   len[r] from T2                                  (32 MiB of table, once per sort)
//   radix_finish_plan_kernel      run starts, the longest run, the decision       (64 workgroups)
//   radix_finish_sort_kernel      step 2                                         (a workgroup of 256 .. 1024 threads per run)
#pragma once

#include "radix_pair_passes.hpp"

namespace glu_hip
{
constexpr uint32_t kFinishRuns = 65536; // runs = values of the 16 top-bit key bits

// lengths[e * 256 + d] = #keys with first top-bit digit d and second top-bit digit e: T2 rows (d, b) summed over the leader's nb
// blocks.  One workgroup per d; thread (g, q) adds word q (counters e = 2q, 2q + 1) of the rows b = g, g + 8, ...
__global__ __launch_bounds__(1024) void radix_finish_lengths_kernel(const uint32_t* __restrict__ t2, uint32_t nb,
                                                                    uint32_t* __restrict__ lengths, const PassPlan* plan,
                                                                    uint32_t pass)
{
    if (plan->off[pass] || plan->skip[pass] == kSkipWithoutCounting) return; // no tables (kernel-uniform)
    __shared__ uint32_t part[8][kPairRadix];
    const uint32_t tid = threadIdx.x, g = tid >> 7, q = tid & 127u, d = blockIdx.x;
    uint32_t lo = 0, hi = 0;
#pragma unroll 4
    for (uint32_t b = g; b < nb; b += 8)
    {
        const uint32_t w = t2[((size_t) d * nb + b) * kPairRowWords + q];
        lo += w & 0xFFFFu;
        hi += w >> 16;
    }
    part[g][2 * q] = lo;
    part[g][2 * q + 1] = hi;
    __syncthreads();
    if (tid < kPairRadix)
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) c += part[k][tid];
        lengths[tid * kPairRadix + d] = c;
    }
}

// starts[r] = exclusive scan of lengths (starts[65536] = n), and the decision: the sort ends in LDS if the lengths are
// exact (they add up to n: a 16-bit counter of T2 that overflowed loses 65536) and the longest run fits the tile of one of the
// in-LDS pass's geometries whose launches follow (numbered geo_first .. geo_last, finish_geometry_capacity; the host enqueues the
// one that suits uniformly drawn keys of this count and the next larger ones: keys that leave some runs empty and the others
// longer -- 31-bit keys, mild skew -- still end in LDS, in a larger tile).  PassPlan::finish = the geometry chosen.
//   accepted: the ordinary passes [first_ordinary, first_ordinary + num_ordinary) are switched off;
//   refused:  the two top-bit passes `pass`, `pass + 1` are switched off (the leader has counted already: its scatter
//             sees skip = kSkipWithoutCounting, which leaves the arrays' roles as they are).
// hint: see glu_radix_sort_s::finish_hint.  64 workgroups, each scans 1024 runs; every workgroup reads all 65536 lengths
// (256 KiB, from L2) for the sum in front of its runs, the total and the longest run, so each reaches the same decision
// without a second launch.
constexpr uint32_t kFinishPlanBlocks = kFinishRuns / 1024;
// tile geometries of the in-LDS pass: 1 = 256 threads x 6 pairs, 2 = 256 x 10, 3 = 256 x 18 (39 KiB of LDS: four workgroups per
// CU), 4 = 512 x 18 (78 KiB: two per CU, the largest that still overlaps one run's memory time with another's ranking)
constexpr uint32_t kFinishGeometries = 4;
__host__ __device__ constexpr uint32_t finish_geometry_capacity(uint32_t g)
{
    return g == 1 ? 256u * 6u : g == 2 ? 256u * 10u : g == 3 ? 256u * 18u : g == 4 ? 512u * 18u : 0u;
}
// the smallest of the enqueued tile geometries [geo_first, geo_last] that holds the longest run (0: none does)
__host__ __device__ inline uint32_t finish_geometry_choice(uint32_t longest, uint32_t geo_first, uint32_t geo_last)
{
    uint32_t geo = 0;
    for (uint32_t g = geo_last; g >= geo_first && g >= 1; g--)
        if (longest <= finish_geometry_capacity(g)) geo = g;
    return geo;
}
__global__ __launch_bounds__(1024) void radix_finish_plan_kernel(const uint32_t* __restrict__ lengths, uint32_t* __restrict__ starts,
                                                                 uint32_t n, uint32_t geo_first, uint32_t geo_last, PassPlan* plan,
                                                                 uint32_t pass,
                                                                 uint32_t first_ordinary, uint32_t num_ordinary,
                                                                 uint32_t* hint, uint32_t attempt, uint32_t top_bit,
                                                                 uint32_t key_bits)
{
    __shared__ uint32_t tmp[3][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const bool tables = !(plan->off[pass] || plan->skip[pass] == kSkipWithoutCounting); // (kernel-uniform)
    uint32_t before = 0, all = 0, longest = 0, mine = 0;
    if (tables)
    {
#pragma unroll 8
        for (uint32_t j = 0; j < kFinishPlanBlocks; j++)
        {
            const uint32_t v = lengths[j * 1024u + tid];
            all += v;
            before += j < b ? v : 0u;
            longest = max(longest, v);
            mine = j == b ? v : mine;
        }
    }
    uint32_t wtotal;
    uint32_t excl = wave_exclusive_sum(mine, lane, wtotal);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        before += __shfl_xor(before, o);
        all += __shfl_xor(all, o);
        longest = max(longest, (uint32_t) __shfl_xor(longest, o));
    }
    if (lane == 0) tmp[0][wave] = before, tmp[1][wave] = all, tmp[2][wave] = longest;
    __shared__ uint32_t wsum[16];
    if (lane == 0) wsum[wave] = wtotal;
    __syncthreads();
    before = all = longest = 0;
#pragma unroll
    for (int w = 0; w < 16; w++)
    {
        before += tmp[0][w];
        all += tmp[1][w];
        longest = max(longest, tmp[2][w]);
        if ((uint32_t) w < wave) excl += wsum[w];
    }
    const uint32_t geo = finish_geometry_choice(longest, geo_first, geo_last);
    // The runs are the values of key bits [top_bit - 16, top_bit): that orders the keys only if no key bit from top_bit up
    // varies -- the host assumed so from what this object's last sort saw, the count kernel of this one has looked
    // (PassPlan::bits_or / bits_nor).  Typed keys and sorts that do not collect the bits are launched with top_bit = key_bits.
    bool range_ok = top_bit >= key_bits;
    uint64_t varying = ~0ull;
    if (plan->bits_valid)
    {
        varying = (uint64_t) (plan->bits_or[0] & plan->bits_nor[0]) | ((uint64_t) (plan->bits_or[1] & plan->bits_nor[1]) << 32);
        if (top_bit < key_bits) range_ok = (varying >> top_bit) == 0;
    }
    const bool accept = tables && all == n && geo != 0 && range_ok;
    if (accept)
    {
        starts[b * 1024u + tid] = before + excl;
        if (b == 0 && tid == 0) starts[kFinishRuns] = n;
    }
    if (b == 0 && tid == 0)
    {
        plan->finish = accept ? geo : 0u;
        plan->finish_longest = tables ? longest : 0xFFFFFFFFu;
        // for the host, which reads it without synchronising: the outcome of attempt number `attempt` (pinned host memory)
        // (attempt << 3 | the geometry chosen, 0 = refused; and which key bits vary, for the next sort's choice of top_bit:
        // words 1, 2, valid for attempt number word 3)
        if (hint)
        {
            if (plan->bits_valid)
            {
                __hip_atomic_store(hint + 1, (uint32_t) varying, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(hint + 2, (uint32_t) (varying >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(hint + 3, attempt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(hint, (attempt << 3) | (accept ? geo : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!accept)
        {
            plan->skip[pass] = kSkipWithoutCounting;
            plan->off[pass + 1] = 1;
        }
    }
    if (b == 0 && accept && tid < num_ordinary) plan->off[first_ordinary + tid] = 1;
}

// The stage of 64-bit keys: only the low 48 key bits differ inside a run (the bits above are the run's, or constant over the
// whole input), so a slot holds 6 + 4 bytes instead of 8 + 4: 512 x 9 pairs stay below 53 KiB and a third workgroup fits the CU.
template<int COUNT, bool VALS>
struct FinishStage48
{
    uint32_t lo[COUNT];
    uint16_t mid[COUNT];
    uint32_t vals[VALS ? COUNT : 1];
    __device__ __forceinline__ void put(uint32_t pos, uint64_t k, uint32_t v)
    {
        lo[pos] = (uint32_t) k;
        mid[pos] = (uint16_t) (k >> 32);
        if (VALS) vals[pos] = v;
    }
    __device__ __forceinline__ void get(uint32_t pos, uint64_t& k, uint32_t& v, uint64_t run_top) const
    {
        k = run_top | ((uint64_t) mid[pos] << 32) | lo[pos];
        v = VALS ? vals[pos] : 0u;
    }
    // (the low 48 bits: all that differs inside a run)
    __device__ __forceinline__ uint64_t key_at(uint32_t pos) const { return ((uint64_t) mid[pos] << 32) | lo[pos]; }
};
template<typename KeyT, int COUNT, bool VALS>
struct FinishStage : PairArray<KeyT, COUNT, VALS>
{
    __device__ __forceinline__ void get(uint32_t pos, KeyT& k, uint32_t& v, KeyT) const { PairArray<KeyT, COUNT, VALS>::get(pos, k, v); }
    __device__ __forceinline__ KeyT key_at(uint32_t pos) const
    {
        KeyT k;
        uint32_t v;
        PairArray<KeyT, COUNT, VALS>::get(pos, k, v);
        return k;
    }
};
template<int COUNT, bool VALS>
struct FinishStage<uint64_t, COUNT, VALS> : FinishStage48<COUNT, VALS>
{
};

template<typename KeyT, int THREADS, int KPT, bool VALS>
struct FinishSmem
{
    static constexpr int RADIX = 256;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    FinishStage<KeyT, TILE, VALS> stage;
    // wave-private running digit counters, 16-bit (a tile has fewer than 65536 slots): with 256 x 18 pairs the workgroup
    // stays below 40 KiB and four of them share a CU (tools/lds_final_pass_bench.hip: 1.06 -> 0.99 ms for 2^28 pairs)
    uint16_t wcnt[WAVES][RADIX];
    uint32_t scan_tmp[WAVES];
    // tie repair (see the kernel): how many tie groups want repairing (their first positions are listed in wcnt, which is idle
    // then), and whether one of them is too long for it
    uint32_t tie_count, tie_bad;
    static constexpr uint32_t TIE_LIST = WAVES * RADIX;
};
static_assert(sizeof(FinishSmem<uint32_t, 256, 18, true>) <= 40 * 1024, "four workgroups per CU");
static_assert(sizeof(FinishSmem<uint32_t, 512, 18, true>) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(FinishSmem<uint64_t, 512, 9, true>) <= 53 * 1024, "64-bit keys: three workgroups per CU");

// the longest run a workgroup of this geometry takes
template<int THREADS, int KPT>
constexpr uint32_t finish_capacity() { return (uint32_t) (THREADS * KPT); }

// A workgroup per run r: pairs [starts[r], starts[r + 1]) of the arrays that hold the data after pass `pass`
// - 1 (PassPlan::flip[pass]) are ordered by key bits [0, low_bits), stably, in place.  Wave-striped items, wave-private
// running digit counters, one scan over (digit, wave), staging in ranked order: the body of radix_sort_single_block_kernel.
// Every wave takes an equal share of the run (a multiple of 64 slots) and ranks only the items its share has -- a run of
// 4096 pairs costs 16 items per lane, not the 18 the longest run needs.  Slots past the run's end hold the key ~0 (largest
// digit in every round, behind every real pair in input order); their loads read the run's last element instead of being
// predicated, so that all loads of a lane are in flight at once.
// 64-bit keys: the same with 8-byte keys in registers and LDS and six rounds for the low 48 bits (low_bits = 48).
// LOOP = false: launched with a workgroup per run (the hardware's dispatcher is the loop over the runs, and the next workgroup
// starts while this one drains its stores: 0.96-0.99 ms for 2^28 pairs where a loop inside the kernel takes 1.05-1.16).  LOOP =
// true: fewer workgroups, each takes every gridDim.x-th run -- for the geometries that are enqueued besides the expected one:
// 65536 workgroups that return at once cost 15-29 us, 8192 cost 5.
// XF: typed keys (signed integers, floats): the first top-bit pass encoded them on load, this pass decodes them on store.
//
// rank_from > 0 (keys with more than 16 bits left to order: 64-bit keys, a segmented sort by 32 bits): the rounds rank only the
// key bits [rank_from, low_bits) -- the TOP of what is left, two rounds instead of six for 64-bit keys -- which orders the run
// except where two keys agree on those bits (a TIE: for uniformly drawn keys a run of 4096 has some 128 tied neighbours on 16
// ranked bits).  Ties are repaired exactly: every position compares itself with its successor in LDS; where both agree on the
// ranked bits and disagree in order on the rest, the lane walks back to the first position of its tie group (a few steps) and
// lists it; after a barrier one lane per listed group sorts the group in place by stable insertion on the whole key.  A group
// longer than kTieMaxGroup, a walk of more than kTieMaxBack steps or more groups than the list holds mark the run as one that
// does not suit this (keys that crowd on the ranked bits): the workgroup then runs ALL rounds, [0, low_bits), on what is staged --
// a stable permutation of the run, so the result is the same.  Groups of EQUAL keys of any length need no repair and cost nothing.
//
// Two callers.  The whole-key sort (plan != nullptr): 65536 runs, the geometry and the arrays come from the PassPlan.  The
// segmented sort (plan == nullptr, radix_seg_passes.hpp: the local sort of the sharded sort): `nruns` runs of (segment, top
// digit of the low bits) in keys_a / vals_a, and the geometry follows from the longest run the runs kernel found (*gate):
// the smallest of [geo_first, geo_last] that holds it; none does: the kernel returns, the ordinary segmented passes run.
template<typename KeyT, int THREADS, int KPT, bool VALS, bool LOOP, bool XF = false>
__global__ __launch_bounds__(THREADS) void radix_finish_sort_kernel(KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b,
                                                                    uint32_t* vals_b, const uint32_t* __restrict__ starts,
                                                                    uint32_t low_bits, const PassPlan* plan, uint32_t pass,
                                                                    uint32_t geometry, uint32_t key_xf = 0,
                                                                    uint32_t nruns = kFinishRuns, const uint32_t* gate = nullptr,
                                                                    uint32_t geo_first = 0, uint32_t geo_last = 0,
                                                                    uint32_t rank_from = 0)
{
    const KeyCodec<KeyT, XF> codec_out(key_xf);
    // (kernel-uniform: the device chose another geometry, or the ordinary passes)
    if (plan ? plan->finish != geometry : finish_geometry_choice(*gate, geo_first, geo_last) != geometry) return;
    using Smem = FinishSmem<KeyT, THREADS, KPT, VALS>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int WQ = WAVES / 4;
    constexpr int SCAN_THREADS = RADIX * WQ;
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    static_assert(WAVES % 4 == 0 && SCAN_THREADS <= THREADS, "offset scan geometry");

    KeyT* keys = plan && plan->flip[pass] ? keys_b : keys_a;
    uint32_t* vals = plan && plan->flip[pass] ? vals_b : vals_a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t run = blockIdx.x; run < nruns; run += LOOP ? gridDim.x : nruns)
    {
    const uint32_t begin = starts[run], end = starts[run + 1];
    const uint32_t len = end - begin;
    if (len == 0 || (!XF && len == 1)) continue; // (workgroup-uniform; a single typed key still has to be decoded)
    // (64-bit keys: the key bits from 48 up, the same for every pair of the run -- and of a pad that comes back from the stage)
    const KeyT run_top = sizeof(KeyT) == 8 ? (KeyT) (keys[begin] & (KeyT) 0xFFFF000000000000ull) : (KeyT) 0;
    const uint32_t share = ((len + WAVES * kWave - 1) / (WAVES * kWave)) * kWave; // slots per wave: <= kWave * KPT
    const uint32_t items = share / kWave;
    const uint32_t wave_off = wave * share + lane;

    KeyT key[KPT];
    uint32_t val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        const bool ok = p < len;
        const uint32_t pc = ok ? p : len - 1;
        const KeyT k = keys[begin + pc];
        const uint32_t v = VALS ? vals[begin + pc] : 0u;
        key[i] = ok ? k : (KeyT) ~(KeyT) 0;
        val[i] = ok ? v : 0u;
    }

    uint16_t* my_cnt = s.wcnt[wave];
    // (workgroup-uniform) the first attempt ranks [rank_from, low_bits) and repairs ties; if the run does not suit that, the
    // second ranks everything
    for (uint32_t shift_begin = rank_from;; shift_begin = 0)
    {
    if (shift_begin != 0 && tid == 0) s.tie_count = 0, s.tie_bad = 0; // (barriers follow before either is used)
    for (uint32_t shift = shift_begin; shift < low_bits; shift += 8)
    {
        constexpr uint32_t MASK = 255u;
        for (int i = tid; i < WAVES * RADIX / 2; i += THREADS) reinterpret_cast<uint32_t*>(&s.wcnt[0][0])[i] = 0;
        __syncthreads();

        uint32_t rank[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if ((uint32_t) i >= items) continue; // (workgroup-uniform)
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            uint16_t* const cnt = my_cnt + d;
            const uint32_t prev = *cnt;
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < 8; bit++)
            {
                int32_t sel; // (as in radix_sort_single_block_kernel)
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = (uint32_t) __popc(plo) + (uint32_t) __popc(phi);
            rank[i] = prev + lower;
            asm volatile("" : "+v"(rank[i]));
            *cnt = (uint16_t) (prev + total);
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
                s.wcnt[sw + 0][sd] = (uint16_t) excl;
                s.wcnt[sw + 1][sd] = (uint16_t) (excl + c0);
                s.wcnt[sw + 2][sd] = (uint16_t) (excl + c0 + c1);
                s.wcnt[sw + 3][sd] = (uint16_t) (excl + c0 + c1 + c2);
            }
        }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if ((uint32_t) i >= items) continue;
            s.stage.put((uint32_t) my_cnt[digit_of<KeyT>(key[i], shift, MASK)] + rank[i], key[i], val[i]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if ((uint32_t) i >= items) continue;
            s.stage.get(wave_off + i * kWave, key[i], val[i], run_top);
        }
        if (shift + 8 < low_bits) __syncthreads();
    }
    if (shift_begin == 0) break;
    // ---- tie repair on the stage (the registers hold a copy of it)
    {
        constexpr uint32_t kTieMaxBack = 16, kTieMaxGroup = 32;
        uint16_t* const list = &s.wcnt[0][0];
        // (only the key bits [0, low_bits) count: a segmented sort by fewer bits than the key has must not look at the others)
        const KeyT low_mask = low_bits >= 8u * sizeof(KeyT) ? (KeyT) ~(KeyT) 0 : (KeyT) ((((KeyT) 1) << low_bits) - 1);
        auto tied = [&](KeyT a, KeyT b) { return (((a ^ b) & low_mask) >> shift_begin) == 0; };
        auto above = [&](KeyT a, KeyT b) { return (a & low_mask) > (b & low_mask); };
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if ((uint32_t) i >= items) continue;
            const uint32_t p = wave_off + i * kWave;
            // (positions from len on are the pads; the key of position p is in key[i], up to the bits above low_bits)
            const bool in = p + 1 < len;
            const KeyT kp = s.stage.key_at(in ? p : 0u), kn = s.stage.key_at(in ? p + 1 : 0u);
            if (in && tied(kp, kn) && above(kp, kn))
            {
                // the first out-of-order neighbours of a tie group list the group's first position
                uint32_t q = p, steps = 0;
                KeyT kq = kp;
                bool mine = true, open = true;
                while (q > 0 && steps < kTieMaxBack)
                {
                    const KeyT kb = s.stage.key_at(q - 1);
                    if (!tied(kb, kq))
                    {
                        open = false;
                        break;
                    }
                    if (above(kb, kq))
                    {
                        mine = false; // (an earlier lane's)
                        break;
                    }
                    kq = kb;
                    q--;
                    steps++;
                }
                if (q == 0) open = false;
                if (mine && open)
                    s.tie_bad = 1u; // the group begins further back than this lane may walk
                else if (mine)
                {
                    const uint32_t at = atomicAdd(&s.tie_count, 1u);
                    if (at < Smem::TIE_LIST) list[at] = (uint16_t) q;
                    else s.tie_bad = 1u;
                }
            }
        }
        __syncthreads();
        const uint32_t groups = s.tie_count;
        if (!s.tie_bad)
        {
            for (uint32_t g = tid; g < groups; g += THREADS)
            {
                const uint32_t first = list[g];
                const KeyT k0 = s.stage.key_at(first);
                uint32_t end = first + 1;
                while (end < len && end - first <= kTieMaxGroup && tied(k0, s.stage.key_at(end))) end++;
                if (end - first > kTieMaxGroup)
                {
                    s.tie_bad = 1u;
                    continue;
                }
                for (uint32_t a = first + 1; a < end; a++) // stable insertion by the whole key
                {
                    KeyT ka;
                    uint32_t va;
                    s.stage.get(a, ka, va, run_top);
                    uint32_t b = a;
                    while (b > first)
                    {
                        KeyT kb;
                        uint32_t vb;
                        s.stage.get(b - 1, kb, vb, run_top);
                        if (!above(kb, ka)) break;
                        s.stage.put(b, kb, vb);
                        b--;
                    }
                    if (b != a) s.stage.put(b, ka, va);
                }
            }
        }
        __syncthreads();
        const bool bad = s.tie_bad != 0u;
        if (bad || groups)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                if ((uint32_t) i >= items) continue;
                s.stage.get(wave_off + i * kWave, key[i], val[i], run_top);
            }
        }
        if (!bad) break;
        __syncthreads(); // (the counters -- the list -- are zeroed by the next round)
    }
    }

#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        if ((uint32_t) i < items && p < len)
        {
            __builtin_nontemporal_store(codec_out.decode(key[i]), &keys[begin + p]);
            if (VALS) __builtin_nontemporal_store(val[i], &vals[begin + p]);
        }
    }
    if (LOOP) __syncthreads(); // (the stage and the counters are reused)
    }
}

} // namespace glu_hip
