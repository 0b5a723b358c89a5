// radix_seg_passes.hpp -- count and scan kernels of SEGMENTED counting passes (glu_radix_sort_run_segments_ptr: the local
// sort of the sharded sort after its exchange, glu_dist_impl.hpp).
//
// A segmented pass is the stable counting pass of the reference (k_radix_sort_counting_shader + BlellochScan +
// k_radix_sort_reordering_shader, glu/RadixSort.hpp:11-183) applied to every SEGMENT of the array on its own, all segments
// in one launch sequence: dst = segment start + (keys of the segment with a smaller digit) + (keys with this digit earlier
// in the segment) -- RadixSort.hpp:174-177 with the "global offset" taken per segment.
//
// The input is cut into SUB-BLOCKS (host: seg_build_image in glu_hip.hip): element ranges that lie inside one piece of one
// segment and inside one workgroup's share of the work.  Sub-blocks are numbered segment-major, in the stable order of the
// segment's elements, so a segment's sub-blocks are the contiguous index range [seg_list[g], seg_list[g + 1]).
//   radix_seg_count_kernel   a workgroup per share, sub-block by sub-block: table[i][d] = #keys of sub-block i with digit d   ([sub-block][digit])
//   radix_seg_scan_kernel    one workgroup per segment:   table[i][d] = absolute destination of the first such key
//   radix_scatter_lines_kernel<..., SEG = true> (radix_scatter_lines.hpp): a workgroup runs the pass body over its sub-blocks
#pragma once

#include "radix_sort_kernels.hpp"

namespace glu_hip
{
// A segmented sort that ENDS IN LDS (glu_hip.hip, seg_run_plan): its first pass is the one on the TOP digit of the bits to sort
// by; after it the array is a sequence of runs (segment, top digit), whose starts are the first table row of every segment
// after that pass's scan.  radix_seg_runs_kernel writes them out and notes the longest run in a device word, the GATE: when it
// fits an LDS tile (gate_cap: the capacity of the largest tile geometry that is enqueued) the pass's scatter and the in-LDS pass
// over the runs (radix_finish_sort_kernel, radix_lds_finish.hpp) run and the ordinary passes, enqueued behind, return at once;
// otherwise the other way round.  One launch sequence either way, decided on the device (as in radix_lds_finish.hpp).
constexpr uint32_t kSegGateNone = 0, kSegGateIfFits = 1, kSegGateIfNot = 2;
__device__ __forceinline__ bool seg_gate_closed(const uint32_t* gate, uint32_t gate_cap, uint32_t gate_mode)
{
    if (gate_mode == kSegGateNone) return false;
    const bool fits = *gate <= gate_cap;
    return gate_mode == kSegGateIfFits ? !fits : fits;
}

// starts[g * RADIX + d] = where the keys of segment g with top digit d begin after the pass (table: the scanned rows of that
// pass, seg_list / seg_start: radix_seg_scan_kernel's), starts[nseg * RADIX] = end; *gate = the longest run.  One workgroup.
template<int RADIX>
__global__ __launch_bounds__(1024) void radix_seg_runs_kernel(const uint32_t* __restrict__ table, const uint32_t* __restrict__ seg_list,
                                                              const uint32_t* __restrict__ seg_start, uint32_t nseg, uint32_t end,
                                                              uint32_t* __restrict__ starts, uint32_t* __restrict__ gate)
{
    __shared__ uint32_t wave_max[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nruns = nseg * RADIX;
    auto start_of = [&](uint32_t i) -> uint32_t {
        if (i >= nruns) return end;
        const uint32_t g = i / RADIX, d = i % RADIX;
        const uint32_t row = seg_list[g];
        return row == seg_list[g + 1] ? seg_start[g] : table[(size_t) row * RADIX + d]; // (a segment without elements has no rows)
    };
    uint32_t longest = 0;
#pragma unroll 8 // (two dependent loads per run: eight runs' chains in flight instead of one -- 2^18 runs are 256 iterations per thread)
    for (uint32_t i = tid; i < nruns; i += 1024)
    {
        const uint32_t a = start_of(i), b = start_of(i + 1);
        starts[i] = a;
        longest = max(longest, b - a);
    }
    if (tid == 0) starts[nruns] = end;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t) __shfl_xor(longest, o));
    if (lane == 0) wave_max[wave] = longest;
    __syncthreads();
    if (tid == 0)
    {
        for (int w = 1; w < 16; w++) longest = max(longest, wave_max[w]);
        *gate = longest;
    }
}

// K1 per sub-block.  Reads 4 B per key; a range may start and end at any element of a 16-byte aligned array.  Workgroup w
// counts the sub-blocks [seg_first[w], seg_first[w + 1]) one after the other -- the same equal share of the elements that
// workgroup w of the scatter kernel moves.  (One workgroup per sub-block, the first version, ran 125 us where the plain
// count kernel takes 86 us for the same 2^27 keys: sub-blocks come in all sizes, two share a CU, and the CU that draws two
// large ones sets the kernel time.)
// BITS_OUT (round 6, the first pass over the long runs of a whole-key sort): also the OR and the AND of every sub-block's keys, for
// radix_seg_scan_kernel to find the segments whose keys agree on every bit the passes order -- a long run of ONE key value, what
// duplicate-heavy inputs are made of -- and take them out of both passes.
template<typename KeyT, int BITS, int THREADS, bool BITS_OUT = false>
__global__ __launch_bounds__(THREADS) void radix_seg_count_kernel(const KeyT* __restrict__ keys,
                                                                  const uint2* __restrict__ subs,
                                                                  const uint32_t* __restrict__ seg_first,
                                                                  uint32_t* __restrict__ table, uint32_t shift, uint32_t mask,
                                                                  const uint32_t* gate = nullptr, uint32_t gate_cap = 0,
                                                                  uint32_t gate_mode = kSegGateNone,
                                                                  const KeyT* __restrict__ keys_alt = nullptr,
                                                                  const PassPlan* plan = nullptr, uint32_t flip_pass = 0, uint32_t swap = 0,
                                                                  KeyT* __restrict__ sub_or = nullptr, KeyT* __restrict__ sub_and = nullptr,
                                                                  const uint32_t* seg_shares = nullptr)
{
    if (seg_gate_closed(gate, gate_cap, gate_mode)) return; // (kernel-uniform)
    // (the long runs of a whole-key sort that ends in LDS: which pair of arrays holds the data is known on the device only)
    if (plan && ((plan->flip[flip_pass] ^ swap) & 1u)) keys = keys_alt;
    __shared__ KeyT bits_tmp[2][THREADS / kWave];
    constexpr int RADIX = 1 << BITS;
    constexpr int WAVES = THREADS / kWave;
    constexpr uint32_t EPV = 16 / sizeof(KeyT); // keys per 16-byte piece (8-byte keys: round 6, the long runs of 64-bit sorts)
    __shared__ uint32_t hist[WAVES][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* my_hist = hist[wave];
    auto dig = [&](KeyT k) { return digit_of<KeyT>(k, shift, mask); };
    // every lane of the wave is active when this runs (wave_tally: one add per group of equal digits when they are few)
    TallyRun run;
    auto add_count = [&](uint32_t dv, uint32_t c) { atomicAdd(&my_hist[dv], c); };
    auto tally = [&](auto peel, uint32_t d) { wave_tally_mode(peel, d, lane, run, add_count); };
    auto key_at = [&](const uint4& a, int e) -> KeyT {
        if constexpr (sizeof(KeyT) == 4) return (KeyT) (e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w);
        else return (KeyT) (e == 0 ? ((uint64_t) a.x | ((uint64_t) a.y << 32)) : ((uint64_t) a.z | ((uint64_t) a.w << 32)));
    };
    KeyT acc_or = 0, acc_and = (KeyT) ~(KeyT) 0;
    auto note = [&](KeyT k) {
        if (BITS_OUT) acc_or |= k, acc_and &= k;
    };
    auto tally_vec = [&](auto peel, const uint4& a) {
#pragma unroll
        for (int e = 0; e < (int) EPV; e++)
        {
            note(key_at(a, e));
            tally(peel, dig(key_at(a, e)));
        }
    };
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    // (seg_shares, round 6, the long runs of a whole-key sort: more shares than workgroups, workgroup w takes the shares w, w +
    // gridDim.x, ... -- the sub-blocks that are left once the runs of one key value have been emptied are then spread over the
    // chip instead of sitting in a few workgroups' shares; how many: decided on the device, radix_finish_long_runs_kernel; nullptr: a
    // share per workgroup)
    const uint32_t nshares = seg_shares ? *seg_shares : gridDim.x;
    for (uint32_t sh = blockIdx.x; sh < nshares; sh += gridDim.x)
    {
    const uint32_t sb_first = seg_first[sh], sb_last = seg_first[sh + 1];
    for (uint32_t sb = sb_first; sb < sb_last; sb++)
    {
        const uint2 r = subs[sb];
        const uint64_t begin = r.x, end = r.y;
        // head: the elements in front of the first 16-byte boundary
        uint64_t vstart = (begin + (EPV - 1)) & ~(uint64_t) (EPV - 1);
        if (vstart > end) vstart = end;
        if (BITS_OUT) acc_or = 0, acc_and = (KeyT) ~(KeyT) 0;
        if (begin + tid < vstart)
        {
            note(keys[begin + tid]);
            atomicAdd(&my_hist[dig(keys[begin + tid])], 1u);
        }
        const uint64_t nvec = (end - vstart) / EPV;
        const uint4* vkeys = reinterpret_cast<const uint4*>(keys + vstart);
        uint64_t vbase = 0;
        auto main_loop = [&](auto peel) { // (twice, chosen once per wave and sub-block from its first keys: see wave_tally)
            for (; vbase + 4 * THREADS <= nvec; vbase += 4 * THREADS) // block-uniform trip count, 4 x 16 B in flight per lane
            {
                if (wave_tally_gives_up(peel, run)) break; // (peel mode only; the wave comes back in the stateless mode)
                const uint4 a = load_streaming(&vkeys[vbase + tid]);
                const uint4 b = load_streaming(&vkeys[vbase + tid + THREADS]);
                const uint4 c = load_streaming(&vkeys[vbase + tid + 2 * THREADS]);
                const uint4 d = load_streaming(&vkeys[vbase + tid + 3 * THREADS]);
                tally_vec(peel, a);
                tally_vec(peel, b);
                tally_vec(peel, c);
                tally_vec(peel, d);
            }
        };
        if (4 * THREADS <= nvec) wave_tally_dispatch(dig(key_at(vkeys[tid], 0)), lane, run, main_loop, add_count);
        for (uint64_t v = vbase + tid; v < nvec; v += THREADS) // lanes may be inactive: plain atomics
        {
            const uint4 a = vkeys[v];
#pragma unroll
            for (int e = 0; e < (int) EPV; e++)
            {
                note(key_at(a, e));
                atomicAdd(&my_hist[dig(key_at(a, e))], 1u);
            }
        }
        const uint64_t tail = vstart + nvec * EPV + tid;
        if (tail < end)
        {
            note(keys[tail]);
            atomicAdd(&my_hist[dig(keys[tail])], 1u);
        }
        if (BITS_OUT)
        {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
            {
                acc_or |= (KeyT) __shfl_xor((unsigned long long) acc_or, o);
                acc_and &= (KeyT) __shfl_xor((unsigned long long) acc_and, o);
            }
            if (lane == 0) bits_tmp[0][wave] = acc_or, bits_tmp[1][wave] = acc_and;
        }
        __syncthreads();
        // the sub-block's row, and the counters back to zero for the next one
        for (int d = tid; d < RADIX; d += THREADS)
        {
            uint32_t c = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++)
            {
                c += hist[w][d];
                hist[w][d] = 0;
            }
            table[(size_t) sb * RADIX + d] = c;
        }
        if (BITS_OUT && tid == 0)
        {
            KeyT o = 0, a = (KeyT) ~(KeyT) 0;
            for (int w = 0; w < WAVES; w++) o |= bits_tmp[0][w], a &= bits_tmp[1][w];
            sub_or[sb] = o;
            sub_and[sb] = a;
        }
        __syncthreads();
    }
    } // shares
}

// K2 per segment: thread d walks the segment's sub-blocks (rows of the table, RADIX consecutive words each: coalesced),
// turns the counts of digit d into running sums, scans the RADIX digit totals of the segment, and adds segment start +
// digit offset to every entry.  (What BlellochScan + the reorder shader's 16-lane global scan do per pass in the
// reference, BlellochScan.hpp:142-190, RadixSort.hpp:148-152 -- per segment.)
// sub_or / sub_and (round 6, the first pass over the long runs of a whole-key sort; radix_seg_count_kernel<.., BITS_OUT> wrote
// them): a segment whose keys agree on every bit of `ordered` -- the bits the passes order -- is in order as it stands (equal keys
// in input order): its sub-blocks are made EMPTY in the descriptor image (subs[i].y = subs[i].x), and this pass's scatter and
// every kernel of the passes behind it find nothing to do there.  A long run of one key value then costs the one read that found
// that out instead of 2 x 20 bytes per pair.
template<int RADIX, typename KeyT = uint32_t>
__global__ __launch_bounds__(RADIX) void radix_seg_scan_kernel(uint32_t* __restrict__ table,
                                                               const uint32_t* __restrict__ seg_list,
                                                               const uint32_t* __restrict__ seg_start,
                                                               const uint32_t* gate = nullptr, uint32_t gate_cap = 0,
                                                               uint32_t gate_mode = kSegGateNone, const uint32_t* nseg_dev = nullptr,
                                                               uint2* subs = nullptr, const KeyT* __restrict__ sub_or = nullptr,
                                                               const KeyT* __restrict__ sub_and = nullptr, const PassPlan* plan = nullptr,
                                                               uint32_t ordered_bits = 0)
{
    if (seg_gate_closed(gate, gate_cap, gate_mode)) return; // (kernel-uniform)
    constexpr int WAVES = (RADIX + kWave - 1) / kWave;
    __shared__ uint32_t wave_sums[WAVES];
    __shared__ KeyT seg_bits[2][WAVES];
    const uint32_t d = threadIdx.x, lane = d & 63, wave = d >> 6;
    // (the bits the long-run passes order: those below the runs' 16, which the device may have chosen -- PassPlan::top_bit)
    if (plan && plan->top_bit) ordered_bits = plan->top_bit - 16u;
    const KeyT ordered = ordered_bits >= 8u * sizeof(KeyT) ? (KeyT) ~(KeyT) 0 : (KeyT) ((((KeyT) 1) << ordered_bits) - 1);
    // (segments counted on the device -- nseg_dev: the workgroups loop over them; else a workgroup per segment)
    const uint32_t nseg = nseg_dev ? *nseg_dev : gridDim.x;
    for (uint32_t g = blockIdx.x; g < nseg; g += gridDim.x)
    {
    const uint32_t i0 = seg_list[g], i1 = seg_list[g + 1];
    if (sub_or)
    {
        KeyT o = 0, a = (KeyT) ~(KeyT) 0;
        for (uint32_t i = i0 + d; i < i1; i += RADIX) o |= sub_or[i], a &= sub_and[i];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1)
        {
            o |= (KeyT) __shfl_xor((unsigned long long) o, s);
            a &= (KeyT) __shfl_xor((unsigned long long) a, s);
        }
        if (lane == 0) seg_bits[0][wave] = o, seg_bits[1][wave] = a;
        __syncthreads();
        o = 0, a = (KeyT) ~(KeyT) 0;
        for (int w = 0; w < WAVES; w++) o |= seg_bits[0][w], a &= seg_bits[1][w];
        __syncthreads(); // (seg_bits is rewritten for the next segment)
        if (((o ^ a) & ordered) == 0) // (workgroup-uniform) one value in the ordered bits: nothing to do for this segment, ever
        {
            for (uint32_t i = i0 + d; i < i1; i += RADIX) subs[i].y = subs[i].x;
            continue;
        }
    }
    // (16 independent loads in flight, then the stores: a load behind a store that the compiler cannot tell apart from it is not
    // moved up, and one row per load latency made 37 us of a segment of 256 sub-blocks)
    constexpr int kRowsInFlight = 16;
    uint32_t run = 0;
    uint32_t i = i0;
    for (; i + kRowsInFlight <= i1; i += kRowsInFlight)
    {
        uint32_t* row = table + (size_t) i * RADIX + d;
        uint32_t c[kRowsInFlight];
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++) c[k] = row[k * RADIX];
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++)
        {
            row[k * RADIX] = run;
            run += c[k];
        }
    }
    for (; i < i1; i++)
    {
        uint32_t* row = table + (size_t) i * RADIX + d;
        const uint32_t c = row[0];
        row[0] = run;
        run += c;
    }
    uint32_t wtotal;
    uint32_t excl = wave_exclusive_sum(run, lane, wtotal);
    if (lane == 0) wave_sums[wave] = wtotal;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) excl += wave_sums[w];
    const uint32_t add = seg_start[g] + excl;
    for (i = i0; i + kRowsInFlight <= i1; i += kRowsInFlight)
    {
        uint32_t* row = table + (size_t) i * RADIX + d;
        uint32_t c[kRowsInFlight];
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++) c[k] = row[k * RADIX];
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++) row[k * RADIX] = c[k] + add;
    }
    for (; i < i1; i++) table[(size_t) i * RADIX + d] += add;
    __syncthreads(); // (wave_sums is rewritten for the next segment)
    }
}

} // namespace glu_hip
