// radix_lds_plan.hpp -- the DECISION of a sort that ends in LDS (radix_lds_finish.hpp has the whole idea): which 16 key bits make
// the runs, how long every run is, whether and in which tile the sort ends in LDS, and which runs are left to segmented passes.
//
//   radix_sample_top_kernel       which 16 key bits make the runs                 (64 workgroups, 256 KiB of keys)
//   radix_finish_lengths_kernel   len[r] from the leader's two-digit table T2 (+ the carries of its wide rows)
//   radix_finish_plan_kernel      run starts, the longest run, the decision       (64 workgroups)
//   radix_finish_long_runs_kernel the runs longer than the chosen tile, as segments of segmented passes
//
// Launched from glu_sort_passes.hpp (the kernels that are not templates are `static`: both of its translation units hold a copy);
// glu_hip.hip takes the sample kernel and the layout of the long runs from here; the kernels of the in-LDS pass itself are launched
// from glu_sort_finish.hip.
#pragma once

#include "radix_lds_finish.hpp"
#include "radix_pair_passes.hpp"

namespace glu_hip
{
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

// long_ok: runs LONGER than the chosen tile do not refuse the sort -- they are segments for segmented counting passes over just
// their elements (radix_finish_long_runs_kernel builds the descriptors on the device, radix_seg_passes.hpp the passes; round 5:
// 4-byte untyped keys with values, round 6: every key type), the in-LDS pass leaves them alone.  The tile is then the smallest
// enqueued one that leaves at most kLongRunsMax runs and an eighth of the pairs to those passes; failing that the largest one, if
// what the attempt then moves is no more than the ordinary passes would (the cost rule in radix_finish_plan_kernel); failing that
// the sort is refused (keys crowded into few runs of few varying bytes are better off with the ordinary passes and their
// skipping of constant digits).
constexpr uint32_t kLongRunsMax = 8192;
constexpr uint32_t kLongRunsMinShare = 4096;  // pairs: the least one workgroup of the segmented passes is given (radix_finish_long_runs_kernel)
// shares per workgroup of the segmented passes: the long runs laid end to end are cut into this many times the number of workgroups,
// workgroup w works on the shares w, w + nwg, ...  With one share each, the runs of ONE key value -- emptied by the first pass's
// scan kernel -- left whole workgroups idle and the few runs of two or three values to a few: 1000 distinct values at 2^28, seven
// runs of two values, 0.37 ms per segmented scatter on seven workgroups (64-bit keys: six such passes).
constexpr uint32_t kLongRunsSharesPerWg = 8;

// The runs longer than the tile the plan chose, as SEGMENTS of segmented passes (radix_seg_passes.hpp): the descriptor image
// the host builds for the sharded sort's local sort (seg_build_image in glu_hip.hip), built on the device from the run starts.
// The long runs laid end to end are cut into nwg equal shares, a sub-block is the part of one run inside one share:
//   image + 0:          subs[kLongRunsMax + nwg] (begin, end) element ranges, in the order of the runs  (nwg here and below: the number
//                       of SHARES, kLongRunsSharesPerWg times the number of workgroups of the segmented passes)
//   image + off_first:  seg_first[nwg + 1]       first sub-block of every workgroup's share
//   image + off_list:   seg_list[kLongRunsMax + 1]  first sub-block of every long run
//   image + off_start:  seg_start[kLongRunsMax]     where the run starts (it stays where it is)
//   hdr[0] = number of long runs (0: none, or the sort does not end in LDS: the segmented kernels return at once), hdr[1] = sub-blocks,
//   hdr[2] = pairs in long runs, hdr[3] = the number of shares the long runs are cut into (seg_first has that many + 1 entries in use)
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
__device__ __forceinline__ void finish_list_long_runs(const uint32_t* starts, uint32_t geo, uint32_t finish_longest, uint32_t nwg /* shares at most */,
                                                      uint32_t* __restrict__ image, uint32_t* __restrict__ hdr, uint32_t (&wsum)[2][16])
{
    const LongRunsLayout lay(nwg);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cap = finish_geometry_capacity(geo);
    constexpr uint32_t PER = kFinishRuns / 1024;
    auto ld = [&](uint32_t i) { return starts[i]; };
    if (geo == 0 || finish_longest <= cap) // (workgroup-uniform) refused, or no run outgrows the tile: nothing to list
    {
        if (tid < 4) hdr[tid] = tid < 3 ? 0u : nwg;
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
        if (tid == 0) hdr[1] = 0u, hdr[3] = nwg;
        return;
    }
    // (a share is what one workgroup of a segmented pass works on; a few long runs of a few thousand pairs cut 256 ways made 256
    // sub-blocks of a hundred pairs, every one a row of 256 counts for the scan kernel to walk: 0.01 % zeros cost 0.15 ms more
    // than 0 %.  Shares behind the last pair are empty.)
    // How many shares: `nwg` is the most there may be, kLongRunsSharesPerWg per workgroup of the segmented passes.  Many shares
    // spread what is left of the long runs, once those of one key value have been emptied, over the chip (1000 distinct values, seven
    // runs of two values each: 0.37 -> 0.05 ms per segmented scatter; 64-bit keys 6.9 -> 4.6 ms) -- but every share boundary
    // is a sub-block more in some run, a row more for the scan kernel's walk along that run and a partial tile more: one run of a
    // tenth of the input cut 2048 ways cost 0.45 ms.  So: as many shares per workgroup as leave the LONGEST run about 64 sub-blocks.
    const uint32_t wgs = nwg / kLongRunsSharesPerWg;
    const uint32_t per_wg = (uint32_t) min<uint64_t>(kLongRunsSharesPerWg, max<uint64_t>(1ull, (uint64_t) total * 64ull / ((uint64_t) finish_longest * wgs)));
    nwg = wgs * per_wg;
    if (tid == 0) hdr[3] = nwg;
    const uint32_t share = max((total + nwg - 1) / nwg, kLongRunsMinShare);
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
static __global__ __launch_bounds__(1024) void radix_finish_lengths_kernel(const uint32_t* __restrict__ t2, uint32_t nb,
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

static __global__ __launch_bounds__(1024) void radix_finish_plan_kernel(const uint32_t* __restrict__ lengths, uint32_t* starts,
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
        // (all 64 loads of a thread in flight at once: eight at a time made eight load latencies, 18 us of kernel for 256 KiB from L2)
#pragma unroll
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
            // rule, is not moved at all; round 5 refused a sort with more than half of its pairs in long runs --
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
static __global__ __launch_bounds__(1024) void radix_finish_long_runs_kernel(const uint32_t* __restrict__ starts, const PassPlan* plan,
                                                                      uint32_t nwg, uint32_t* __restrict__ image, uint32_t* __restrict__ hdr)
{
    __shared__ uint32_t wsum[2][16];
    finish_list_long_runs(starts, plan->finish, plan->finish_longest, nwg, image, hdr, wsum);
}

} // namespace glu_hip
