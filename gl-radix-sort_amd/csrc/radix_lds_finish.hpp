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
//      of passes (radix_pair_passes.hpp).  `top` is chosen ON THE DEVICE from a sample of the keys (radix_sample_top_kernel: the
//      highest bit that varies, at least 16; typed keys: the key's width).  After them the array is a sequence of 65536 RUNS, run r =
//      the pairs whose key has the value r in those bits, in input order.  The leader's two-digit histogram T2 already holds
//      every run's length: len[r] = sum over blocks b of T2[r & 255][b][r >> 8].
//   2. ONE pass then orders every run by the low top - 16 bits inside LDS, in place, and writes it back where it was: 16 B per pair
//      instead of the 2 x 20.5 of two more passes -- 52.25 B per pair in all (64-bit keys: 80.25 instead of 225).  Since round 6
//      that pass is radix_finish_bucket_kernel (radix_lds_bucket.hpp: one bucket round on unique words, whole 128-byte lines in
//      and out); the ballot-ranked radix_finish_sort_kernel below (rounds of 8 bits of rank / scan / re-stage; 64-bit keys: two
//      rounds on the top 16 of the remaining bits + exact tie repair) takes the runs that one lists as crowded, every run when
//      fewer than nine bits are left to order (PassPlan::finish_rounds), and the runs of a SEGMENTED sort that ends in LDS.
//      Typed keys (signed, float) are encoded on load by the first top-bit pass and decoded on store by the in-LDS pass.
//   3. The decision is made on the device (radix_lds_plan.hpp: radix_finish_plan_kernel, after the leader's count kernel and before
//      its scatter, from exact run lengths -- the leader recounts the rows of T2 whose 16-bit counters wrapped, round 6): the tile
//      is the smallest enqueued geometry that leaves at most 8192 runs and an eighth of the pairs longer than itself; failing
//      that the largest one, if ordering its long runs moves no more bytes than the ordinary passes would.  LONG runs are listed
//      as segments and ordered by segmented passes over just their elements (radix_seg_passes.hpp; every key type since round
//      6), the in-LDS pass leaves them alone.  More than 8192 long runs, long runs that cost more than the ordinary passes, or a
//      key bit above `top` that varies after all refuse the attempt: the two top-bit passes are switched off and the passes of
//      the ordinary sort, enqueued behind every attempt, run instead (PassPlan::off, PassPlan::skip).  One launch sequence
//      either way -- the sort stays asynchronous and graph-capturable; the kernels of the sequence not taken return at once, on
//      a stream of the sort object's own since round 6 (glu_radix_sort_s::side).  A refused attempt costs one read of the keys;
//      its outcome and the key bits that varied reach the host through a pinned word that the next sort call reads without
//      synchronising (glu_radix_sort_s::finish_hint).
//
//   radix_lds_plan.hpp            the sample, the run lengths, the decision, the list of long runs
//   radix_finish_sort_kernel      step 2 by ballot rounds                         (a workgroup of 256 .. 1024 threads per run)
//   radix_finish_ranges_kernel    the same for runs longer than a tile / split runs of a segmented sort
//   radix_lds_bucket.hpp          step 2 by one bucket round (round 6: the kernel a headline sort runs)
#pragma once

#include "radix_sort_kernels.hpp"

namespace glu_hip
{
constexpr uint32_t kFinishRuns = 65536; // runs = values of the 16 top-bit key bits

// The runs that radix_finish_bucket_kernel (radix_lds_bucket.hpp, round 6) found crowded, for radix_finish_sort_kernel behind it:
// kCrowdedLists lists by the run number's low bits -- 65536 workgroups appending to ONE list are 65536 atomics on one address,
// 0.6 ms when every run is crowded -- each with its counter in a 128-byte line of its own:
//   words [k * kCrowdedCountStride]                                            how many runs list k holds
//   words [kCrowdedLists * kCrowdedCountStride + k * capacity + i]              the i-th of them
// The counters are zeroed by radix_finish_plan_kernel.
constexpr uint32_t kCrowdedLists = 256, kCrowdedCountStride = 32;
// (which list: the top bits of run * 0x9E3779B1 -- the run number's low bits put every 16th run, a pattern keys do produce, into 16
// of the 256 lists.  Consecutive run numbers spread evenly under this hash: 254 .. 258 of 65536 per list, twice the mean at worst
// for strided subsets; a list holds twice its mean, and a run that finds its list full takes the next one.)
__host__ __device__ constexpr uint32_t crowded_list_capacity(uint32_t nruns) { return 2u * ((nruns + kCrowdedLists - 1u) / kCrowdedLists); }
__host__ __device__ constexpr size_t crowded_list_words(uint32_t nruns)
{
    return (size_t) kCrowdedLists * kCrowdedCountStride + (size_t) kCrowdedLists * crowded_list_capacity(nruns);
}
__device__ __forceinline__ void crowded_list_append(uint32_t* lists, uint32_t nruns, uint32_t run)
{
    const uint32_t cap = crowded_list_capacity(nruns);
    for (uint32_t k = (run * 0x9E3779B1u) >> 24;; k = (k + 1u) & (kCrowdedLists - 1u))
    {
        const uint32_t i = atomicAdd(&lists[k * kCrowdedCountStride], 1u);
        if (i < cap)
        {
            lists[kCrowdedLists * kCrowdedCountStride + k * cap + i] = run;
            return;
        }
        // (full -- its count stays above the capacity, the reader clamps it -- the next list; all lists together hold 2 x nruns)
    }
}

// The plan (radix_finish_plan_kernel below): starts[r] = exclusive scan of the run lengths (starts[65536] = n), and the decision:
// the sort ends in LDS if the lengths are exact (they add up to n: a 16-bit counter of T2 that overflowed loses 65536) and the
// longest run fits the tile of one of the in-LDS pass's geometries whose launches follow (numbered geo_first .. geo_last,
// finish_geometry_capacity; the host enqueues the one that suits uniformly drawn keys of this count and the next larger ones:
// keys that leave some runs empty and the others longer -- 31-bit keys, mild skew -- still end in LDS, in a larger tile).
// PassPlan::finish = the geometry chosen.
//   accepted: the ordinary passes [first_ordinary, first_ordinary + num_ordinary) are switched off;
//   refused:  the two top-bit passes `pass`, `pass + 1` are switched off (the leader has counted already: its scatter
//             sees skip = kSkipWithoutCounting, which leaves the arrays' roles as they are).
// hint: see glu_radix_sort_s::finish_hint.
constexpr uint32_t kFinishPlanBlocks = kFinishRuns / 1024;
// tile geometries of the in-LDS pass: 1 = 256 threads x 6 pairs, 2 = 256 x 10, 3 = 256 x 18 (39 KiB of LDS: four workgroups per
// CU), 4 = 512 x 18 (78 KiB: two per CU, the largest that still overlaps one run's memory time with another's ranking)
constexpr uint32_t kFinishGeometries = 4;
// (segmented sorts only, round 5) 5 = 1024 x 17: 147 KiB, ONE workgroup per CU -- nothing overlaps one run's memory time there, but
// the runs of the sharded sort at eight ranks (16384 pairs) fit it whole
constexpr uint32_t kSegFinishGeometries = 5;
__host__ __device__ constexpr uint32_t finish_geometry_capacity(uint32_t g)
{
    return g == 1 ? 256u * 6u : g == 2 ? 256u * 10u : g == 3 ? 256u * 18u : g == 4 ? 512u * 18u : g == 5 ? 1024u * 17u : 0u;
}
// the smallest of the enqueued tile geometries [geo_first, geo_last] that holds the longest run (0: none does)
__host__ __device__ inline uint32_t finish_geometry_choice(uint32_t longest, uint32_t geo_first, uint32_t geo_last)
{
    uint32_t geo = 0;
    for (uint32_t g = geo_last; g >= geo_first && g >= 1; g--)
        if (longest <= finish_geometry_capacity(g)) geo = g;
    return geo;
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
static_assert(sizeof(FinishSmem<uint32_t, 1024, 17, true>) <= 160 * 1024 - 256, "one workgroup per CU");
static_assert(sizeof(FinishSmem<uint64_t, 512, 9, true>) <= 53 * 1024, "64-bit keys: three workgroups per CU");

// the longest run a workgroup of this geometry takes
template<int THREADS, int KPT>
constexpr uint32_t finish_capacity() { return (uint32_t) (THREADS * KPT); }

// The rounds of the in-LDS pass on what a workgroup holds in registers: `items` wave-striped items per lane (key[i], val[i] =
// slot wave_off + i * 64 of the tile; slots from `len` on are pads: key ~0, behind every real pair), ordered stably by the key
// bits [0, low_bits) -- on return key[i], val[i] are slot wave_off + i * 64 of the ORDERED tile.  Rounds of 8 bits of rank /
// scan / re-stage as in radix_sort_single_block_kernel: ballot ranking against wave-private digit counters, one scan over
// (digit, wave), staging in ranked order.
//
// rank_from > 0 (64-bit keys: up to 48 bits are left to order.  4-byte keys -- a segmented sort by 32 bits leaves 24 -- run every
// round instead, TIES = false below, whatever rank_from their launch passes): the rounds rank only the
// key bits [rank_from, low_bits) -- the TOP of what is left, two rounds instead of six for 64-bit keys -- which orders the run
// except where two keys agree on those bits (a TIE: for uniformly drawn keys a run of 4096 has some 128 tied neighbours on 16
// ranked bits).  Ties are repaired exactly: every position compares itself with its successor in LDS; where both agree on the
// ranked bits and disagree in order on the rest, the lane walks back to the first position of its tie group (a few steps) and
// lists it; after a barrier one lane per listed group sorts the group in place by stable insertion on the whole key.  A group
// longer than kTieMaxGroup, a walk of more than kTieMaxBack steps or more groups than the list holds mark the run as one that
// does not suit this (keys that crowd on the ranked bits): the workgroup then runs ALL rounds, [0, low_bits), on what is staged --
// a stable permutation of the run, so the result is the same.  Groups of EQUAL keys of any length need no repair and cost nothing.
// Phase clock of the in-LDS pass (tools/finish_stamps_bench.hip builds the STAMPS = true instantiation; the library does not):
// s_memtime at phase boundaries, summed per phase by the first and the last wave of every workgroup into the workgroup's own 16
// words of `out` (plain stores: atomics on shared words slow the very loads that are being timed).
//   0 load issue -> keys and values in registers (first barrier)   1 ranking (ballots)   2 offsets scan
//   3 staging (put, barrier, get)   4 tie detection + repair   5 store issue   6 (spare)   7 whole workgroup
template<bool STAMPS>
struct FinishClock
{
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0, t0 = 0;
    __device__ __forceinline__ void start()
    {
        if (STAMPS) tprev = t0 = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void stamp(int slot)
    {
        if (STAMPS)
        {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            acc[slot] += t - tprev;
            tprev = t;
        }
    }
    __device__ __forceinline__ void flush(unsigned long long* out, uint32_t lane, uint32_t wave, uint32_t waves)
    {
        if (STAMPS && out && lane == 0 && (wave == 0 || wave == waves - 1))
        {
            acc[7] = __builtin_amdgcn_s_memtime() - t0;
#pragma unroll
            for (int i = 0; i < 8; i++) out[(size_t) blockIdx.x * 16 + (wave == 0 ? 0 : 8) + i] = acc[i];
        }
    }
};

// TIES = false (4-byte keys: at most 24 bits are ever left, and the repair's registers would cost the 256 x 18 tile its fourth
// workgroup per CU): every round runs, rank_from is not looked at.
template<typename KeyT, int THREADS, int KPT, bool VALS, bool TIES = (sizeof(KeyT) == 8), bool STAMPS = false>
__device__ __forceinline__ void finish_rank_rounds(FinishSmem<KeyT, THREADS, KPT, VALS>& s, KeyT (&key)[KPT], uint32_t (&val)[KPT],
                                                   uint32_t items, uint32_t len, uint32_t wave_off, uint32_t low_bits,
                                                   uint32_t rank_from_arg, KeyT run_top, uint32_t tid, uint32_t lane, uint32_t wave,
                                                   FinishClock<STAMPS>& clock)
{
    const uint32_t rank_from = TIES ? rank_from_arg : 0u;
    using Smem = FinishSmem<KeyT, THREADS, KPT, VALS>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    // the offsets scan: a thread per (digit, group of GW waves' counters); GW = 4 where the waves come in fours, else all
    constexpr int GW = WAVES % 4 == 0 ? 4 : WAVES;
    constexpr int WQ = WAVES / GW;
    constexpr int SCAN_THREADS = RADIX * WQ;
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    static_assert(SCAN_THREADS <= THREADS, "offset scan geometry");
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
        clock.stamp(shift == shift_begin && shift_begin == rank_from ? 0 : 3);

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
        clock.stamp(1);

        {
            const uint32_t sd = tid / WQ, sw = (tid % WQ) * GW;
            uint32_t c[GW];
            uint32_t csum = 0;
#pragma unroll
            for (int g = 0; g < GW; g++)
            {
                c[g] = tid < SCAN_THREADS ? (uint32_t) s.wcnt[sw + g][sd] : 0u;
                csum += c[g];
            }
            uint32_t excl = 0;
            if (wave < SCAN_WAVES)
            {
                uint32_t wtotal;
                excl = wave_exclusive_sum(csum, lane, wtotal);
                if (SCAN_WAVES > 1 && lane == 0) s.scan_tmp[wave] = wtotal;
            }
            if (SCAN_WAVES > 1)
            {
                __syncthreads();
                excl += sum_of_preceding_waves(s.scan_tmp, SCAN_WAVES, wave, lane);
            }
            if (tid < SCAN_THREADS)
            {
#pragma unroll
                for (int g = 0; g < GW; g++)
                {
                    s.wcnt[sw + g][sd] = (uint16_t) excl;
                    excl += c[g];
                }
            }
        }
        __syncthreads();
        clock.stamp(2);

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
    clock.stamp(3);
    if (!TIES || shift_begin == 0) break;
    // ---- tie repair on the stage (the registers hold a copy of it)
    if constexpr (TIES)
    {
        constexpr uint32_t kTieMaxBack = 16, kTieMaxGroup = 32;
        uint16_t* const list = &s.wcnt[0][0];
        // (only the key bits [0, low_bits) count: a segmented sort by fewer bits than the key has must not look at the others)
        const KeyT low_mask = low_bits >= 8u * sizeof(KeyT) ? (KeyT) ~(KeyT) 0 : (KeyT) ((((KeyT) 1) << low_bits) - 1);
        auto tied = [&](KeyT a, KeyT b) { return (((a ^ b) & low_mask) >> shift_begin) == 0; };
        auto above = [&](KeyT a, KeyT b) { return (a & low_mask) > (b & low_mask); };
        // Every position against its successor, from REGISTERS: position p = wave_off + i * 64 is item i of this lane, its
        // successor is item i of the next lane -- for lane 63 item i + 1 of lane 0, and behind the wave's last item the first slot
        // of the next wave's share, the one key that comes from the stage.  (Reading both keys of every position from the stage
        // made this phase a third of the pass: profiles/r05/finish_stamps_u64_rank16_before.txt.)
        const KeyT next_wave_first = s.stage.key_at(min(wave_off - lane + items * kWave, (uint32_t) Smem::TILE - 1u));
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if ((uint32_t) i >= items) continue;
            const uint32_t p = wave_off + i * kWave;
            // (positions from len on are the pads)
            const bool in = p + 1 < len;
            const KeyT kp = key[i];
            KeyT kn = (KeyT) __shfl_down((unsigned long long) key[i], 1);
            {
                KeyT first_of_next = next_wave_first;
                if (i + 1 < KPT)
                    if ((uint32_t) (i + 1) < items) first_of_next = (KeyT) __shfl((unsigned long long) key[i + 1 < KPT ? i + 1 : i], 0);
                if (lane == 63) kn = first_of_next;
            }
            // out-of-order neighbours inside a tie group: their position is listed (one LDS atomic per wave and item: the wave's
            // count; every lane's slot follows from the ballot)
            const bool inv = in && tied(kp, kn) && above(kp, kn);
            const uint64_t invs = __ballot(inv);
            if (invs) // (wave-uniform)
            {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&s.tie_count, (uint32_t) __popcll(invs));
                at = (uint32_t) __builtin_amdgcn_readfirstlane((int) at) +
                     __builtin_amdgcn_mbcnt_hi((uint32_t) (invs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) invs, 0u));
                if (inv)
                {
                    if (at < Smem::TIE_LIST) list[at] = (uint16_t) p;
                    else s.tie_bad = 1u;
                }
            }
        }
        __syncthreads();
        const uint32_t listed = min(s.tie_count, Smem::TIE_LIST);
        if (!s.tie_bad)
        {
            // a lane per listed position: it walks back to the first position of its tie group -- unless an earlier pair of the
            // group is out of order too (that one's lane repairs the group) -- finds the group's end and sorts the group in place
            // by stable insertion on the whole key
            for (uint32_t g = tid; g < listed; g += THREADS)
            {
                const uint32_t p = list[g];
                uint32_t first = p, steps = 0;
                KeyT kq = s.stage.key_at(p);
                bool mine = true, open = p > 0;
                while (first > 0 && steps < kTieMaxBack)
                {
                    const KeyT kb = s.stage.key_at(first - 1);
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
                    first--;
                    steps++;
                }
                if (first == 0) open = false;
                if (mine && open) s.tie_bad = 1u; // the group begins further back than a lane may walk
                // (who repairs which group is settled before any group is touched: a lane that walked back through a group
                // that is being repaired could take it for its own)
                list[g] = mine && !open ? (uint16_t) first : (uint16_t) 0xFFFFu;
            }
        }
        __syncthreads();
        if (!s.tie_bad)
        {
            for (uint32_t g = tid; g < listed; g += THREADS)
            {
                const uint32_t first = list[g];
                if (first == 0xFFFFu) continue;
                const KeyT k0 = s.stage.key_at(first);
                uint32_t end = first + 2; // (the group has an out-of-order pair: at least two positions)
                while (end < len && end - first <= kTieMaxGroup && tied(k0, s.stage.key_at(end))) end++;
                if (end - first > kTieMaxGroup)
                {
                    s.tie_bad = 1u;
                    continue;
                }
                for (uint32_t a = first + 1; a < end; a++)
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
        const uint32_t groups = listed;
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
        clock.stamp(4);
        if (!bad) break;
        __syncthreads(); // (the counters -- the list -- are zeroed by the next round)
    }
    }
}

// A workgroup per run r: the pairs [starts[r], starts[r + 1]) are ordered by key bits [0, low_bits), stably.  Every wave takes
// an equal share of the run (a multiple of 64 slots) and ranks only the items its share has -- a run of 4096 pairs costs 16
// items per lane, not the 18 the longest run needs.  Slots past the run's end hold the key ~0 (largest digit in every round,
// behind every real pair in input order); their loads read the run's last element instead of being predicated, so that all
// loads of a lane are in flight at once.  64-bit keys: 8-byte keys in registers, 6 + 4 bytes per slot in LDS.
// LOOP = false: launched with a workgroup per run (the hardware's dispatcher is the loop over the runs, and the next workgroup
// starts while this one drains its stores: 0.96-0.99 ms for 2^28 pairs where a loop inside the kernel takes 1.05-1.16).  LOOP =
// true: fewer workgroups, each takes every gridDim.x-th run -- for the geometries that are enqueued besides the expected one:
// 65536 workgroups that return at once cost 15-29 us, 8192 cost 5.
// XF: typed keys (signed integers, floats): the first top-bit pass encoded them on load, this pass decodes them on store.
//
// Two callers.  The whole-key sort (plan != nullptr): 65536 runs IN PLACE in the arrays that hold the data after pass `pass` - 1
// (PassPlan::flip[pass]); the geometry comes from the PassPlan.  The segmented sort (plan == nullptr, radix_seg_passes.hpp: the
// local sort of the sharded sort): `nruns` runs of (segment, top digit of the low bits) FROM keys_a / vals_a TO keys_b / vals_b,
// if the longest run the runs kernel found (*gate) is at most gate_cap (otherwise the ordinary segmented passes run); runs
// longer than this geometry's tile are left to radix_finish_ranges_kernel.
#ifndef GLU_FINISH_U64_WAVES
#define GLU_FINISH_U64_WAVES 6 // (tuning builds: 1 = no such request)
#endif
#ifndef GLU_FINISH_KERNEL_ATTR
#define GLU_FINISH_KERNEL_ATTR // (tuning builds of tools/finish_stamps_bench.hip: e.g. __attribute__((amdgpu_waves_per_eu(6))))
#endif
// (8-byte keys, 512 x 9: the 53 KiB stage lets three workgroups share a CU, which takes at most 80 VGPRs: the second launch bound
// asks the compiler for six waves per SIMD -- it needs 81 without it)
template<typename KeyT, int THREADS, int KPT, bool VALS, bool LOOP, bool XF = false, bool STAMPS = false>
__global__ __launch_bounds__(THREADS, (sizeof(KeyT) == 8 && THREADS == 512 ? GLU_FINISH_U64_WAVES : 1)) GLU_FINISH_KERNEL_ATTR void radix_finish_sort_kernel(KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b,
                                                                    uint32_t* vals_b, const uint32_t* __restrict__ starts,
                                                                    uint32_t low_bits, const PassPlan* plan, uint32_t pass,
                                                                    uint32_t geometry, uint32_t key_xf = 0,
                                                                    uint32_t nruns = kFinishRuns, const uint32_t* gate = nullptr,
                                                                    uint32_t gate_cap = 0, uint32_t rank_bits = 16,
                                                                    unsigned long long* stamps = nullptr,
                                                                    const uint32_t* __restrict__ run_list = nullptr,
                                                                    uint32_t len_above = 0)
{
    FinishClock<STAMPS> clock;
    if (plan && plan->top_bit) low_bits = plan->top_bit - 16u; // (the device chose the runs' bits: radix_sample_top_kernel)
    // (more than rank_bits bits left to order: the rounds rank the top rank_bits .. rank_bits + 7 of them, ties are repaired)
    const uint32_t rank_from = low_bits > rank_bits ? ((low_bits - rank_bits) / 8u) * 8u : 0u;
    const KeyCodec<KeyT, XF> codec_out(key_xf);
    // (kernel-uniform: the device chose another geometry, or the ordinary passes)
    // (geometry 0: whichever tile the device chose -- the launches behind radix_finish_bucket_kernel that take the runs it listed,
    // split by the runs' LENGTH: one in the tile the sort is expected to take for the runs that fit it, one in the largest enqueued
    // tile for the runs of more than len_above pairs, i.e. those the first one leaves.  A run is ordered in the smallest of the
    // two tiles that holds it, whatever tile the device chose for the bucket kernel: 2^20 distinct values at 2^28 -- runs of 4096
    // +- 1024 pairs, every one crowded, the 512 x 18 tile chosen -- took 1.56 ms in that tile alone.)
    if (plan ? (geometry ? plan->finish != geometry : plan->finish == 0u) : *gate > gate_cap) return;
    using Smem = FinishSmem<KeyT, THREADS, KPT, VALS>;
    constexpr int WAVES = Smem::WAVES;

    const KeyT* keys = plan && plan->flip[pass] ? keys_b : keys_a;
    const uint32_t* vals = plan && plan->flip[pass] ? vals_b : vals_a;
    KeyT* out_keys = plan ? const_cast<KeyT*>(keys) : keys_b;
    uint32_t* out_vals = plan ? const_cast<uint32_t*>(vals) : vals_b;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (round 6: the whole-key sort's in-LDS pass is radix_finish_bucket_kernel, radix_lds_bucket.hpp; this kernel is launched
    // behind it for the runs that one found crowded and listed: run_list = the crowded lists above)
    // (or for every run, PassPlan::finish_rounds.  Workgroup i works off list i % kCrowdedLists: the grid is a multiple of that.)
    const bool listed = run_list && !plan->finish_rounds;
    if (run_list && !listed && !LOOP) return; // (the launch for all runs is the looping one)
    const uint32_t list = blockIdx.x & (kCrowdedLists - 1u);
    const uint32_t todo = listed ? min(run_list[list * kCrowdedCountStride], crowded_list_capacity(nruns)) : nruns;
    const uint32_t step = !LOOP ? todo : listed ? max(gridDim.x / kCrowdedLists, 1u) : gridDim.x;
    const uint32_t* const my_list = run_list + kCrowdedLists * kCrowdedCountStride + list * crowded_list_capacity(nruns);
    for (uint32_t it = listed ? blockIdx.x / kCrowdedLists : blockIdx.x; it < todo; it += step)
    {
    const uint32_t run = listed ? my_list[it] : it;
    const uint32_t begin = starts[run], end = starts[run + 1];
    const uint32_t len = end - begin;
    if (len == 0 || (plan && !XF && len == 1)) continue; // (workgroup-uniform; a single typed key still has to be decoded)
    // (a run longer than the tile: radix_finish_ranges_kernel's in a segmented sort, the segmented passes' in a whole-key sort)
    if (len > (uint32_t) Smem::TILE) continue;
    // (launched for whichever tile the device chose: the runs longer than THAT tile are the segmented passes' too; the runs of at
    // most len_above pairs are the launch's in the smaller tile)
    if (plan && geometry == 0u && (len > finish_geometry_capacity(plan->finish) || len <= len_above)) continue;
    // (64-bit keys: the key bits from 48 up, the same for every pair of the run -- and of a pad that comes back from the stage)
    const KeyT run_top = sizeof(KeyT) == 8 ? (KeyT) (keys[begin] & (KeyT) 0xFFFF000000000000ull) : (KeyT) 0;
    const uint32_t share = ((len + WAVES * kWave - 1) / (WAVES * kWave)) * kWave; // slots per wave: <= kWave * KPT
    const uint32_t items = share / kWave;
    const uint32_t wave_off = wave * share + lane;

    clock.start();
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

    finish_rank_rounds<KeyT, THREADS, KPT, VALS, (sizeof(KeyT) == 8), STAMPS>(s, key, val, items, len, wave_off, low_bits, rank_from, run_top,
                                                                             tid, lane, wave, clock);

#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        if ((uint32_t) i < items && p < len)
        {
            __builtin_nontemporal_store(codec_out.decode(key[i]), &out_keys[begin + p]);
            if (VALS) __builtin_nontemporal_store(val[i], &out_vals[begin + p]);
        }
    }
    clock.stamp(5);
    if (LOOP) __syncthreads(); // (the stage and the counters are reused)
    }
    clock.flush(stamps, lane, wave, WAVES);
}

// The general form of the in-LDS pass, OUT OF PLACE (src -> dst), for what the kernel above does not take: runs longer than a
// tile, and runs that are SPLIT over several workgroups.
//
// An item of work is (run r, part j of 2^split_log2): the pairs of run r whose key bits [0, low_bits) lie in the j-th of
// 2^split_log2 equal ranges of that key space, [klo, khi) -- the sharded sort at eight ranks has runs of 16384 pairs, four
// workgroups take a quarter of the key range each.  The workgroup walks its range in pieces [lo, hi) that fit its tile:
//   * COUNT: every wave streams over its contiguous share of the run (from L2: the workgroups of a run are neighbours in launch
//     order on one XCD) and counts the pairs whose key lies in [lo, hi) -- ballot, popcount, no barrier inside the stream; one
//     barrier for the waves' counts;
//   * more than TILE: hi moves down (in proportion, at least halving towards lo + 1) and the count is repeated.  A single key value
//     that alone outgrows the tile is copied through in input order (equal keys are in order already);
//   * at most TILE: PLACE -- a second stream appends those pairs to the stage in input order (every wave from its own first
//     position) --, the staged pairs are ordered by finish_rank_rounds and stored at the cursor: the run's start + the number
//     of the run's pairs below klo (counted by the first stream) + what the workgroup has stored so far.
// So any run of any key distribution comes out right; what it costs is reads of the run from L2, once per piece.
// min_len: runs of at most this many pairs are the kernel above's (0: every run is this kernel's).
// done (whole-key sort, in place otherwise): done[r] = 1 tells the copy-back kernel that run r's result is in dst.
template<typename KeyT, int THREADS, int KPT, bool VALS, bool XF = false>
__global__ __launch_bounds__(THREADS) void radix_finish_ranges_kernel(const KeyT* __restrict__ src_keys, const uint32_t* __restrict__ src_vals,
                                                                      KeyT* __restrict__ dst_keys, uint32_t* __restrict__ dst_vals,
                                                                      const uint32_t* __restrict__ starts, uint32_t nruns,
                                                                      uint32_t low_bits, uint32_t rank_from, uint32_t split_log2,
                                                                      uint32_t min_len, const uint32_t* gate, uint32_t gate_cap,
                                                                      uint32_t key_xf = 0)
{
    const KeyCodec<KeyT, XF> codec_out(key_xf);
    if (gate && *gate > gate_cap) return; // (kernel-uniform: the ordinary passes run)
    using Smem = FinishSmem<KeyT, THREADS, KPT, VALS>;
    constexpr int WAVES = Smem::WAVES;
    constexpr uint32_t TILE = (uint32_t) Smem::TILE;
    constexpr int CK = 8;                  // keys per lane and chunk of the gather
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    __shared__ uint32_t gcnt[2][WAVES]; // the waves' counts of a stream: pairs in the range, pairs below the workgroup's range
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t key_space = low_bits >= 64u ? ~0ull : (1ull << low_bits); // (low_bits <= 48 for 64-bit keys)
    const uint64_t low_mask = key_space - 1ull;
    const uint32_t parts = 1u << split_log2;
    // items in launch order: the parts of one run are 8 workgroups apart, i.e. neighbours on one XCD (workgroups are dealt to
    // the eight XCDs round-robin), so that one of them brings the run into that XCD's L2 and the others find it there
    const uint32_t runs8 = (nruns + 7u) & ~7u;
    const uint64_t nitems = (uint64_t) runs8 * parts;
    for (uint64_t item = blockIdx.x; item < nitems; item += gridDim.x)
    {
    const uint32_t xcd = (uint32_t) (item & 7u);
    const uint64_t in_xcd = item >> 3;
    const uint32_t part = (uint32_t) (in_xcd & (parts - 1u));
    const uint32_t run = (uint32_t) (in_xcd >> split_log2) * 8u + xcd;
    if (run >= nruns) continue;
    const uint32_t begin = starts[run], end = starts[run + 1];
    const uint32_t len = end - begin;
    if (len == 0 || len <= min_len) continue; // (workgroup-uniform)
    const KeyT run_top = sizeof(KeyT) == 8 ? (KeyT) (src_keys[begin] & (KeyT) 0xFFFF000000000000ull) : (KeyT) 0;
    const uint64_t klo = split_log2 ? (uint64_t) part << (low_bits - split_log2) : 0ull;
    const uint64_t khi = split_log2 ? (uint64_t) (part + 1u) << (low_bits - split_log2) : key_space;
    uint32_t below_klo = 0; // pairs of the run below the workgroup's key range (known after the first count)

    // Every wave streams over ITS contiguous share of the run (a multiple of 64 slots), batches of CK keys per lane in flight at
    // once; no barrier inside the streams, so the waves of the CU hide each other's load latency.
    const uint32_t wshare = ((len + WAVES * kWave - 1) / (WAVES * kWave)) * kWave;
    const uint32_t wbegin = wave * wshare;
    const uint32_t batches = (wshare + kWave * CK - 1) / (kWave * CK);
    auto load_keys = [&](uint32_t b, KeyT (&k)[CK]) {
#pragma unroll
        for (int i = 0; i < CK; i++)
        {
            const uint32_t p = wbegin + b * (kWave * CK) + i * kWave + lane;
            k[i] = src_keys[begin + (p < len ? p : len - 1u)];
        }
    };
    auto load_vals = [&](uint32_t b, uint32_t (&v)[CK]) {
#pragma unroll
        for (int i = 0; i < CK; i++)
        {
            const uint32_t p = wbegin + b * (kWave * CK) + i * kWave + lane;
            v[i] = VALS ? src_vals[begin + (p < len ? p : len - 1u)] : 0u;
        }
    };
    auto in_share = [&](uint32_t b, int i) { // (is slot i of batch b of this lane a pair of the run and of this wave's share?)
        const uint32_t off = b * (kWave * CK) + i * kWave + lane;
        return off < wshare && wbegin + off < len;
    };
    // COUNT: how many pairs of the run have lo <= key bits < hi (returned: all waves'; wbase: those of the waves before this
    // one); count_below: also how many lie below klo (into below_klo).  One barrier.
    uint32_t wbase = 0;
    // (a key lies in [lo, hi) iff (key bits - lo) <= hi - lo - 1 in the key's own unsigned arithmetic: one compare, no 64-bit
    // arithmetic for 4-byte keys)
    const KeyT kmask = (KeyT) low_mask, klo_k = (KeyT) klo;
    auto count_range = [&](uint64_t lo, uint64_t hi, bool count_below) -> uint32_t {
        const KeyT lo_k = (KeyT) lo, span_k = (KeyT) (hi - lo - 1ull);
        uint32_t wcount = 0, below = 0;
        for (uint32_t b = 0; b < batches; b++)
        {
            KeyT cur[CK];
            load_keys(b, cur);
#pragma unroll
            for (int i = 0; i < CK; i++)
            {
                const KeyT kl = cur[i] & kmask;
                const bool ok = in_share(b, i);
                wcount += (uint32_t) __popcll(__ballot(ok && (KeyT) (kl - lo_k) <= span_k));
                if (count_below) below += (uint32_t) __popcll(__ballot(ok && kl < klo_k));
            }
        }
        if (lane == 0) gcnt[0][wave] = wcount, gcnt[1][wave] = below;
        __syncthreads();
        uint32_t total = 0;
        wbase = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++)
        {
            const uint32_t c = gcnt[0][w];
            wbase += (uint32_t) w < wave ? c : 0u;
            total += c;
            if (count_below) below_klo += gcnt[1][w];
        }
        __syncthreads(); // (gcnt is rewritten by the next count)
        // (workgroup-uniform values that come out of LDS or of vector arithmetic: told to the compiler, so that the loops around
        // this stay scalar control flow)
        if (count_below) below_klo = (uint32_t) __builtin_amdgcn_readfirstlane((int) below_klo);
        return (uint32_t) __builtin_amdgcn_readfirstlane((int) total);
    };
    // PLACE: the pairs counted by the last count_range go to the stage in input order (positions wbase .. of this wave), or
    // -- through -- straight to dst at `out` (a single key value: they are in order).  One barrier at the end.
    auto place_range = [&](uint64_t lo, uint64_t hi, bool through, uint32_t out) {
        const KeyT lo_k = (KeyT) lo, span_k = (KeyT) (hi - lo - 1ull);
        uint32_t pos0 = wbase;
        for (uint32_t b = 0; b < batches; b++)
        {
            KeyT cur[CK];
            uint32_t vcur[CK];
            load_keys(b, cur);
            load_vals(b, vcur);
#pragma unroll
            for (int i = 0; i < CK; i++)
            {
                const bool in = in_share(b, i) && (KeyT) ((cur[i] & kmask) - lo_k) <= span_k;
                const uint64_t bal = __ballot(in);
                const uint32_t pos = pos0 + __builtin_amdgcn_mbcnt_hi((uint32_t) (bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) bal, 0u));
                if (in)
                {
                    if (through)
                    {
                        dst_keys[out + pos] = codec_out.decode(cur[i]);
                        if (VALS) dst_vals[out + pos] = vcur[i];
                    }
                    else
                        s.stage.put(pos, cur[i], vcur[i]);
                }
                pos0 += (uint32_t) __popcll(bal);
            }
        }
        __syncthreads(); // the stage is complete
    };

    uint32_t cursor = begin;
    bool first = true;
    for (uint64_t lo = klo; lo < khi;)
    {
        uint64_t hi = khi;
        uint32_t c;
        for (;;)
        {
            c = count_range(lo, hi, first && split_log2 != 0);
            if (first) cursor = begin + below_klo;
            first = false;
            if (c <= TILE || hi - lo == 1ull) break;
            // too many for the tile: a smaller piece (in proportion to what was found, with a margin; at least one key value)
            const uint64_t span = hi - lo;
            uint64_t next = (uint64_t) ((double) span * (0.85 * (double) TILE / (double) c));
            next = next < 1ull ? 1ull : next;
            next = next > span / 2ull + (span & 1ull) ? span / 2ull + (span & 1ull) : next;
            next = (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) next) |
                   ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (next >> 32)) << 32); // (uniform)
            hi = lo + next;
        }
        if (c > TILE)
            place_range(lo, hi, true, cursor); // one key value, more pairs than a tile: they are in order -- copied through
        else if (c > 0)
        {
            place_range(lo, hi, false, 0u);
            const uint32_t share = ((c + WAVES * kWave - 1) / (WAVES * kWave)) * kWave;
            const uint32_t items = share / kWave;
            const uint32_t wave_off = wave * share + lane;
            KeyT key[KPT];
            uint32_t val[KPT];
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t p = wave_off + i * kWave;
                key[i] = (KeyT) ~(KeyT) 0;
                val[i] = 0u;
                if ((uint32_t) i < items && p < c) s.stage.get(p, key[i], val[i], run_top);
            }
            __syncthreads(); // (the stage is rewritten by the rounds)
            FinishClock<false> clock;
            finish_rank_rounds<KeyT, THREADS, KPT, VALS>(s, key, val, items, c, wave_off, low_bits, rank_from, run_top, tid, lane, wave, clock);
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t p = wave_off + i * kWave;
                if ((uint32_t) i < items && p < c)
                {
                    __builtin_nontemporal_store(codec_out.decode(key[i]), &dst_keys[cursor + p]);
                    if (VALS) __builtin_nontemporal_store(val[i], &dst_vals[cursor + p]);
                }
            }
        }
        cursor += c;
        lo = hi;
        __syncthreads(); // (the stage, the counters and the chunk counts are reused)
    }
    }
}

} // namespace glu_hip
