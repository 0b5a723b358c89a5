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
//   3. The decision is made on the device (radix_finish_plan_kernel, after the leader's count kernel and before its scatter, from
//      exact run lengths): the tile is the smallest enqueued geometry that leaves at most 8192 runs and an eighth of the pairs
//      longer than itself; those LONG runs are listed as segments and ordered by two segmented passes over just their elements
//      (radix_seg_passes.hpp; 4-byte untyped keys with values), the in-LDS pass leaves them alone.  Keys that crowd into few runs
//      (more than half of the pairs in long runs), a wrapped 16-bit counter of T2, or a key bit above `top` that varies after all
//      refuse the attempt: the two top-bit passes are switched off and the passes of the ordinary sort, enqueued behind every
//      attempt, run instead (PassPlan::off, PassPlan::skip).  One launch sequence either way -- the sort stays asynchronous and
//      graph-capturable; the kernels of the sequence not taken return at once, on a stream of the sort object's own since round
//      6 (glu_radix_sort_s::side).  A refused attempt costs one read of the keys; its outcome and the key bits that varied reach
//      the host through a pinned word that the next sort call reads without synchronising (glu_radix_sort_s::finish_hint).
//
//   radix_sample_top_kernel       which 16 key bits make the runs                 (64 workgroups, 256 KiB of keys)
//   radix_finish_lengths_kernel   len[r] from T2                                  (32 MiB of table, once per sort)
//   radix_finish_plan_kernel      run starts, the longest run, the decision       (64 workgroups)
//   radix_finish_long_runs_kernel the runs longer than the chosen tile, as segments of segmented passes
//   radix_finish_sort_kernel      step 2 by ballot rounds                         (a workgroup of 256 .. 1024 threads per run)
//   radix_finish_ranges_kernel    the same for runs longer than a tile / split runs of a segmented sort
#pragma once

#include "radix_pair_passes.hpp"

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
// Which 16 key bits make the runs of a sort that ends in LDS?  The top 16 of the bits that VARY: keys below 2^28 make 4096 runs
// of the whole key's top 16 bits (sixteen times too long) and 65536 of bits [12, 28).  Which bits vary is known exactly only
// after the keys have been read (the leader's count kernel collects them), and that kernel must know its digit before it reads:
// so a few workgroups look at a SAMPLE first -- 16384 16-byte pieces spread evenly over the array, every thread one load, all in flight
// at once (one workgroup walking 65536 pieces took 122 us: a TLB miss per load, in series) --
// and writes PassPlan::top_bit = the highest bit that varies in the sample + 1 (at least 16, at least `floor_top`, at most the
// key's width; 64-bit keys: moved up to 40 or 48 where a digit would straddle the two key words) and the shift every kernel of
// the two top-bit passes subtracts.  A bit above it that varies after all (a rare key the sample missed) is seen by the exact
// collection: the plan kernel then refuses, the ordinary passes run, and the host hands the exact top bit to the next sort as
// floor_top.  (Round 4 guessed from the object's previous sort: the first sort of small-range keys was always the refused one.)
constexpr uint32_t kSampleTopBlocks = 64; // x 256 threads x one 16-byte piece: 256 KiB of keys, all loads in flight at once
template<typename KeyT>
__global__ __launch_bounds__(256) void radix_sample_top_kernel(const KeyT* __restrict__ keys, uint32_t n, uint32_t key_bits,
                                                               uint32_t floor_top, PassPlan* plan)
{
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ uint32_t red[4][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t W = sizeof(KeyT) / 4; // words per key
    const uint64_t nvec = (uint64_t) n * sizeof(KeyT) / 16;
    const uint32_t total = gridDim.x * 256u;
    const uint32_t samples = (uint32_t) (nvec < total ? nvec : total);
    const uint64_t stride = samples ? nvec / samples : 1;
    uint32_t o[2] = {0, 0}, no[2] = {0, 0}; // OR of the low / high key words seen, OR of their complements
    const uint32_t j = blockIdx.x * 256u + tid;
    if (j < samples)
    {
        const u32x4_t x = reinterpret_cast<const u32x4_t*>(keys)[(uint64_t) j * stride];
        if (W == 1)
        {
            o[0] = x.x | x.y | x.z | x.w;
            no[0] = ~x.x | ~x.y | ~x.z | ~x.w;
        }
        else
        {
            o[0] = x.x | x.z;
            no[0] = ~x.x | ~x.z;
            o[1] = x.y | x.w;
            no[1] = ~x.y | ~x.w;
        }
    }
    // (the first and the last key: constant arrays with one odd key at either end are a classic)
    if (blockIdx.x == 0 && tid == 0 && n)
    {
        const KeyT f = keys[0], l = keys[n - 1];
        o[0] |= (uint32_t) f | (uint32_t) l;
        no[0] |= ~(uint32_t) f | ~(uint32_t) l;
        if (W == 2)
        {
            o[1] |= (uint32_t) ((uint64_t) f >> 32) | (uint32_t) ((uint64_t) l >> 32);
            no[1] |= ~(uint32_t) ((uint64_t) f >> 32) | ~(uint32_t) ((uint64_t) l >> 32);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
    {
        o[0] |= __shfl_xor(o[0], s);
        o[1] |= __shfl_xor(o[1], s);
        no[0] |= __shfl_xor(no[0], s);
        no[1] |= __shfl_xor(no[1], s);
    }
    if (lane == 0) red[0][wave] = o[0], red[1][wave] = o[1], red[2][wave] = no[0], red[3][wave] = no[1];
    __syncthreads();
    if (tid == 0)
    {
        for (int w = 1; w < 4; w++) o[0] |= red[0][w], o[1] |= red[1][w], no[0] |= red[2][w], no[1] |= red[3][w];
        // the workgroups' words meet in the plan (agent-scope atomics); the last one to arrive draws the conclusion
        __hip_atomic_fetch_or(&plan->sample_or[0], o[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(&plan->sample_nor[0], no[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (W == 2)
        {
            __hip_atomic_fetch_or(&plan->sample_or[1], o[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_or(&plan->sample_nor[1], no[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t done = __hip_atomic_fetch_add(&plan->sample_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done + 1 == gridDim.x)
        {
            const uint32_t vo0 = __hip_atomic_load(&plan->sample_or[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t vn0 = __hip_atomic_load(&plan->sample_nor[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t vo1 = W == 2 ? __hip_atomic_load(&plan->sample_or[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t vn1 = W == 2 ? __hip_atomic_load(&plan->sample_nor[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint64_t varying = (uint64_t) (vo0 & vn0) | ((uint64_t) (vo1 & vn1) << 32);
            uint32_t top = varying ? 64u - (uint32_t) __builtin_clzll(varying) : 0u;
            top = max(max(top, floor_top), 16u);
            top = min(top, key_bits);
            if (W == 2 && top > 32 && top < 48 && top != 40) top = top < 40 ? 40u : 48u; // a digit stays inside one key word
            plan->top_bit = top;
            plan->shift_down[0] = plan->shift_down[1] = key_bits - top;
        }
    }
}

// long_ok (round 5: 4-byte untyped keys with values): runs LONGER than the chosen tile do not refuse the sort any more -- they
// are segments for two segmented counting passes over just their elements (radix_finish_long_runs_kernel builds the
// descriptors on the device, radix_seg_passes.hpp the passes), the in-LDS pass leaves them alone.  The tile is then the smallest
// enqueued one that leaves at most kLongRunsMax runs and an eighth of the pairs to those passes; failing that the largest one,
// if it leaves at most half of the pairs; failing that the sort is refused as before (keys crowded into few runs are better off
// with the ordinary passes and their skipping of constant digits).
constexpr uint32_t kLongRunsMax = 8192;

// The runs longer than the tile the plan chose, as SEGMENTS of segmented passes (radix_seg_passes.hpp): the descriptor image
// the host builds for the sharded sort's local sort (seg_build_image in glu_hip.hip), built on the device from the run starts.
// The long runs laid end to end are cut into nwg equal shares, a sub-block is the part of one run inside one share:
//   image + 0:          subs[kLongRunsMax + nwg] (begin, end) element ranges, in the order of the runs
//   image + off_first:  seg_first[nwg + 1]       first sub-block of every workgroup's share
//   image + off_list:   seg_list[kLongRunsMax + 1]  first sub-block of every long run
//   image + off_start:  seg_start[kLongRunsMax]     where the run starts (it stays where it is)
//   hdr[0] = number of long runs (0: none, or the sort does not end in LDS: the segmented kernels return at once), hdr[1] = sub-blocks,
//   hdr[2] = pairs in long runs
// One workgroup (radix_finish_long_runs_kernel, behind the plan kernel); thread t owns the runs [64 t, 64 t + 64).
struct LongRunsLayout
{
    uint32_t nwg, off_first, off_list, off_start, words;
    __host__ __device__ explicit LongRunsLayout(uint32_t nwg_) : nwg(nwg_)
    {
        off_first = 2u * (kLongRunsMax + nwg);
        off_list = off_first + nwg + 1u;
        off_start = off_list + kLongRunsMax + 1u;
        words = off_start + kLongRunsMax;
    }
};
__device__ __forceinline__ void finish_list_long_runs(const uint32_t* starts, uint32_t geo, uint32_t finish_longest, uint32_t nwg,
                                                      uint32_t* __restrict__ image, uint32_t* __restrict__ hdr, uint32_t (&wsum)[2][16])
{
    const LongRunsLayout lay(nwg);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cap = finish_geometry_capacity(geo);
    constexpr uint32_t PER = kFinishRuns / 1024;
    auto ld = [&](uint32_t i) { return starts[i]; };
    if (geo == 0 || finish_longest <= cap) // (workgroup-uniform) refused, or no run outgrows the tile: nothing to list
    {
        if (tid < 3) hdr[tid] = 0u;
        return;
    }
    // (every workgroup's share starts empty; the sub-block that begins at a share's first position fills it in)
    for (uint32_t w = tid; w <= nwg; w += 1024) image[lay.off_first + w] = 0xFFFFFFFFu;
    uint32_t cnt = 0, len = 0;
    {
        uint32_t a = ld(tid * PER);
        for (uint32_t j = 0; j < PER; j++)
        {
            const uint32_t e = ld(tid * PER + j + 1);
            if (e - a > cap) cnt++, len += e - a;
            a = e;
        }
    }
    auto block_exclusive = [&](uint32_t v, int slot, uint32_t& total) -> uint32_t {
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(v, lane, wtotal);
        __syncthreads(); // (the slot's last readers are done)
        if (lane == 0) wsum[slot][wave] = wtotal;
        __syncthreads();
        total = 0;
        for (uint32_t w = 0; w < 16; w++)
        {
            excl += w < wave ? wsum[slot][w] : 0u;
            total += wsum[slot][w];
        }
        return excl;
    };
    uint32_t nseg, total;
    uint32_t seg = block_exclusive(cnt, 0, nseg);
    uint32_t pos = block_exclusive(len, 1, total);
    // (the plan allowed at most kLongRunsMax long runs; a sort that did not ask for this has none that are not refused)
    const bool active = nseg != 0 && nseg <= kLongRunsMax;
    if (tid == 0)
    {
        hdr[0] = active ? nseg : 0u;
        hdr[2] = active ? total : 0u;
    }
    if (!active) // (workgroup-uniform)
    {
        if (tid == 0) hdr[1] = 0u;
        return;
    }
    const uint32_t share = (total + nwg - 1) / nwg; // >= cap / nwg > 0
    // sub-blocks of this thread's long runs, then their numbers
    uint32_t subs = 0;
    {
        uint32_t a = ld(tid * PER), p = pos;
        for (uint32_t j = 0; j < PER; j++)
        {
            const uint32_t e = ld(tid * PER + j + 1), l = e - a;
            if (l > cap)
            {
                subs += (p + l - 1) / share - p / share + 1;
                p += l;
            }
            a = e;
        }
    }
    uint32_t nsb;
    uint32_t sb = block_exclusive(subs, 0, nsb);
    {
        uint32_t a = ld(tid * PER), p = pos, g = seg;
        for (uint32_t j = 0; j < PER; j++)
        {
            const uint32_t e = ld(tid * PER + j + 1), l = e - a;
            if (l > cap)
            {
                image[lay.off_list + g] = sb;
                image[lay.off_start + g] = a;
                for (uint32_t w = p / share; w <= (p + l - 1) / share; w++)
                {
                    const uint32_t b0 = max(p, w * share), b1 = min(p + l, (w + 1) * share);
                    image[2 * sb] = a + (b0 - p);
                    image[2 * sb + 1] = a + (b1 - p);
                    if (b0 == w * share) image[lay.off_first + w] = sb; // (this sub-block begins workgroup w's share)
                    sb++;
                }
                p += l;
                g++;
            }
            a = e;
        }
    }
    __syncthreads();
    if (tid == 0)
    {
        hdr[1] = nsb;
        image[lay.off_list + nseg] = nsb;
    }
    // shares behind the last pair (total < nwg * share) are empty: they begin and end at nsb
    for (uint32_t w = tid; w <= nwg; w += 1024)
        if (image[lay.off_first + w] == 0xFFFFFFFFu) image[lay.off_first + w] = nsb;
}

// lengths[e * 256 + d] = #keys with first top-bit digit d and second top-bit digit e: T2 rows (d, b) summed over the leader's nb
// blocks.  One workgroup per d; thread (g, q) adds word q (counters e = 2q, 2q + 1) of the rows b = g, g + 8, ...
// wide (round 6, radix_pair_passes.hpp): the counts >> 16 of the rows whose 16-bit counters wrapped, per block -- the lengths are exact
// whatever share of the input one key value holds.
__global__ __launch_bounds__(1024) void radix_finish_lengths_kernel(const uint32_t* __restrict__ t2, uint32_t nb,
                                                                    uint32_t* __restrict__ lengths, const PassPlan* plan,
                                                                    uint32_t pass, const uint32_t* __restrict__ wide = nullptr)
{
    if (plan->off[pass] || plan->skip[pass] == kSkipWithoutCounting) return; // no tables (kernel-uniform)
    __shared__ uint32_t part[8][kPairRadix];
    __shared__ uint32_t wide_list[1024], wide_n; // (block, wide row) pairs of this digit value: b << 8 | k
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
    if (tid == 0) wide_n = 0;
    __syncthreads();
    if (wide && tid < nb) // (thread b: has block b a wide row of this digit value?)
    {
        const uint32_t* hdr = wide + (size_t) tid * kPairWideStride;
        const uint32_t nw = hdr[0] <= kPairWideRows ? hdr[0] : 0u; // (~0: more rows than the block could put right -- the lengths will not add up)
        for (uint32_t k = 0; k < nw; k++)
            if (hdr[1 + k] == d) wide_list[atomicAdd(&wide_n, 1u)] = (tid << 8) | k;
    }
    __syncthreads();
    if (tid < kPairRadix)
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) c += part[k][tid];
        for (uint32_t i = 0; i < wide_n; i++)
        {
            const uint32_t bk = wide_list[i];
            c += wide[(size_t) (bk >> 8) * kPairWideStride + 16 + (bk & 255u) * kPairRadix + tid] << 16;
        }
        lengths[tid * kPairRadix + d] = c;
    }
}

__global__ __launch_bounds__(1024) void radix_finish_plan_kernel(const uint32_t* __restrict__ lengths, uint32_t* starts,
                                                                 uint32_t n, uint32_t geo_first, uint32_t geo_last, PassPlan* plan,
                                                                 uint32_t pass,
                                                                 uint32_t first_ordinary, uint32_t num_ordinary,
                                                                 uint32_t* hint, uint32_t attempt, uint32_t top_bit,
                                                                 uint32_t key_bits, uint32_t long_ok, uint32_t* crowded_lists,
                                                                 uint32_t* outcomes = nullptr, uint32_t pair_bytes = 8)
{
    __shared__ uint32_t tmp[3][16];
    __shared__ uint32_t over_tmp[2][kFinishGeometries][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const bool tables = !(plan->off[pass] || plan->skip[pass] == kSkipWithoutCounting); // (kernel-uniform)
    uint32_t before = 0, all = 0, longest = 0, mine = 0;
    uint32_t over_len[kFinishGeometries] = {}, over_cnt[kFinishGeometries] = {}; // per tile geometry: pairs in / number of longer runs
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
            if (long_ok && b == 0) // (workgroup 0 makes the decision for all: plan->finish)
            {
#pragma unroll
                for (uint32_t g = 0; g < kFinishGeometries; g++)
                {
                    const bool over = v > finish_geometry_capacity(g + 1);
                    over_len[g] += over ? v : 0u;
                    over_cnt[g] += over ? 1u : 0u;
                }
            }
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
    if (long_ok && b == 0)
    {
#pragma unroll
        for (uint32_t g = 0; g < kFinishGeometries; g++)
        {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
            {
                over_len[g] += __shfl_xor(over_len[g], o);
                over_cnt[g] += __shfl_xor(over_cnt[g], o);
            }
            if (lane == 0) over_tmp[0][g][wave] = over_len[g], over_tmp[1][g][wave] = over_cnt[g];
        }
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
    // which key bits vary (exact: the leader's count kernel has looked at every key; typed keys: not collected, all of them may)
    if (plan->top_bit) top_bit = plan->top_bit; // (chosen on the device from a sample of the keys: radix_sample_top_kernel)
    uint64_t varying = ~0ull;
    if (plan->bits_valid)
        varying = (uint64_t) (plan->bits_or[0] & plan->bits_nor[0]) | ((uint64_t) (plan->bits_or[1] & plan->bits_nor[1]) << 32);
    // What the two ways cost in bytes moved (round 6).  The ordinary sort: a counting pass (a read of the keys + both arrays read
    // and written) for every key BYTE that varies -- the passes on constant bytes are skipped.  The sort that ends in LDS: the
    // leader's read, two counting passes, the in-LDS pass over the short runs, and the segmented passes (2 for 4-byte keys, 6 for
    // 8-byte keys) over the pairs of long runs -- priced as if no long run were one key value (those are not moved at all).
    // Zipf-distributed small integers: 70 B/pair against 60, the ordinary passes win; three values: 76 against 80.
    const uint32_t key_bytes = key_bits / 8u;
    uint32_t varying_bytes = 0;
    for (uint32_t kb = 0; kb < key_bytes; kb++) varying_bytes += ((varying >> (8u * kb)) & 0xFFull) != 0ull ? 1u : 0u;
    const uint64_t pass_bytes = 2ull * pair_bytes + key_bytes;
    const uint64_t ordinary_cost = (uint64_t) n * varying_bytes * pass_bytes;
    auto attempt_cost = [&](uint32_t long_pairs) {
        return (uint64_t) n * (key_bytes + 4ull * pair_bytes) + (uint64_t) (n - long_pairs) * 2ull * pair_bytes +
               (uint64_t) long_pairs * (key_bytes == 4u ? 2ull : 6ull) * pass_bytes;
    };
    uint32_t geo = finish_geometry_choice(longest, geo_first, geo_last);
    if (long_ok && b == 0 && geo != geo_first && geo_first >= 1)
    {
        // (some run outgrows the tile that suits uniform keys: may it, and a few others, go to the segmented passes instead?)
        uint32_t pick = 0;
        for (uint32_t g = geo_last; g >= geo_first; g--)
        {
            uint32_t ol = 0, oc = 0;
            for (int w = 0; w < 16; w++) ol += over_tmp[0][g - 1][w], oc += over_tmp[1][g - 1][w];
            const bool few = oc <= kLongRunsMax && ol <= n / 8u;
            // (round 6: however many pairs the long runs hold -- a long run of one key value, which is what fills long runs as a
            // rule, is not moved at all; round 5 refused the sort when more than half of the pairs sat in long runs --
            // ... as long as that is not more than the ordinary passes would move.)
            const bool tolerable = g == geo_last && oc <= kLongRunsMax && attempt_cost(ol) <= ordinary_cost;
            if (few || (tolerable && pick == 0)) pick = g;
        }
        if (pick) geo = geo == 0 ? pick : min(geo, pick);
    }
    // The runs are the values of key bits [top_bit - 16, top_bit): that orders the keys only if no key bit from top_bit up
    // varies -- the host assumed so from what this object's last sort saw, the count kernel of this one has looked
    // (PassPlan::bits_or / bits_nor).  Typed keys and sorts that do not collect the bits are launched with top_bit = key_bits.
    bool range_ok = top_bit >= key_bits;
    if (plan->bits_valid && top_bit < key_bits) range_ok = (varying >> top_bit) == 0;
    const bool accept = tables && all == n && geo != 0 && range_ok; // (workgroup 0's is the decision: only it knows of long runs)
    // (the run starts are written whatever the decision: nobody reads them unless plan->finish says so)
    starts[b * 1024u + tid] = before + excl;
    if (b == 0 && tid == 0) starts[kFinishRuns] = n;
    if (b == 0 && crowded_lists && tid < kCrowdedLists) crowded_lists[tid * kCrowdedCountStride] = 0u;
    if (b == 0 && tid == 0)
    {
        plan->finish = accept ? geo : 0u;
        plan->finish_longest = tables ? longest : 0xFFFFFFFFu;
        if (outcomes) outcomes[attempt & 255u] = (attempt << 3) | (accept ? geo : 0u); // (glu_radix_sort_read_profile: per sort of a window)
        // which kernel orders the runs: with fewer than nine bits left to order -- or varying, where that is known -- one ballot
        // round beats the bucket round (whose buckets such keys crowd)
        {
            const uint32_t low_bits = top_bit - 16u;
            const uint64_t low_mask = low_bits >= 64u ? ~0ull : (1ull << low_bits) - 1ull;
            const uint32_t to_order = plan->bits_valid ? (uint32_t) __popcll(varying & low_mask) : low_bits;
            plan->finish_rounds = to_order < 9u ? 1u : 0u;
        }
        // for the host, which reads it without synchronising (pinned host memory): the outcome of attempt number `attempt` --
        // attempt << 3 | the geometry chosen, 0 = refused, stored LAST and with release: the host reads it first, then which key
        // bits vary (words 1, 2, valid for attempt number word 3) and the top bit this attempt used (word 4)
        if (hint)
        {
            if (plan->bits_valid)
            {
                __hip_atomic_store(hint + 1, (uint32_t) varying, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(hint + 2, (uint32_t) (varying >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(hint + 3, attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(hint + 4, top_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(hint, (attempt << 3) | (accept ? geo : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!accept)
        {
            plan->skip[pass] = kSkipWithoutCounting;
            plan->off[pass + 1] = 1;
            // (the ordinary passes may start on a stream of their own before the two top-bit scatters have said so)
            plan->flip[pass + 1] = plan->flip[pass + 2] = pass > 0 ? plan->flip[pass] : 0u;
        }
    }
    if (b == 0 && accept && tid < num_ordinary) plan->off[first_ordinary + tid] = 1;
}


// (Folding this into the plan kernel -- its last workgroup to finish -- was measured: the fences of the ticket cost the plan kernel
// 10 us more than this launch does, profiles/r06/last_sort_kernels_2p28_plan_merged.txt.)
__global__ __launch_bounds__(1024) void radix_finish_long_runs_kernel(const uint32_t* __restrict__ starts, const PassPlan* plan,
                                                                      uint32_t nwg, uint32_t* __restrict__ image, uint32_t* __restrict__ hdr)
{
    __shared__ uint32_t wsum[2][16];
    finish_list_long_runs(starts, plan->finish, plan->finish_longest, nwg, image, hdr, wsum);
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
// rank_from > 0 (keys with more than 16 bits left to order: 64-bit keys, a segmented sort by 32 bits): the rounds rank only the
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
                                                                    uint32_t except_geometry = 0)
{
    FinishClock<STAMPS> clock;
    if (plan && plan->top_bit) low_bits = plan->top_bit - 16u; // (the device chose the runs' bits: radix_sample_top_kernel)
    // (more than rank_bits bits left to order: the rounds rank the top rank_bits .. rank_bits + 7 of them, ties are repaired)
    const uint32_t rank_from = low_bits > rank_bits ? ((low_bits - rank_bits) / 8u) * 8u : 0u;
    const KeyCodec<KeyT, XF> codec_out(key_xf);
    // (kernel-uniform: the device chose another geometry, or the ordinary passes)
    // (geometry 0: whichever tile the device chose, except except_geometry -- the launches behind radix_finish_bucket_kernel that
    // take the runs it listed: one in the tile the sort is expected to take, one in the largest enqueued tile for any other choice)
    if (plan ? (geometry ? plan->finish != geometry : (plan->finish == 0u || plan->finish == except_geometry)) : *gate > gate_cap) return;
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
    // (launched for whichever tile the device chose: the runs longer than THAT tile are the segmented passes' too)
    if (plan && geometry == 0u && len > finish_geometry_capacity(plan->finish)) continue;
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
