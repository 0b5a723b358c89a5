// glu_sort_passes.hpp -- the launchers of ONE counting pass of the radix sort (count -> row scan -> scatter; the leader / follower
// of a pair of passes; the decision kernels of a sort that tries to end in LDS behind the leader's count): the implementation
// behind dispatch_pass<KeyT> (glu_sort_object.hpp).  Included by glu_sort_passes_u32.hip and glu_sort_passes_u64.hip only, which
// instantiate it for one key width each -- the two halves of what used to be most of glu_hip.hip's three-minute compile.
#pragma once
#include "glu_sort_object.hpp"
#include "radix_scatter_lines.hpp"
#include "radix_pair_passes.hpp"
#include "radix_lds_plan.hpp"

namespace glu_hip
{
namespace host
{
// XF: this pass encodes keys on load and / or decodes them on store (first / last pass of a typed sort); every other
// pass runs the instantiation without the codec arithmetic.
template<typename KeyT, int BITS, bool LARGE, bool XF, bool VALS>
glu_status launch_pass(glu_radix_sort_s* s, const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v,
                       size_t count, uint32_t shift, uint32_t bits, uint32_t* histogram_out, hipStream_t stream,
                       uint32_t xform = 0, PlanArgs pa = PlanArgs())
{
    using G = GeometryFor<KeyT, BITS, LARGE, VALS>;
    constexpr int RADIX = 1 << BITS;
    const uint32_t tiles = (uint32_t) ((count + G::TILE - 1) / G::TILE);
    uint64_t cap = (uint64_t) usable_cus(s) * G::BLOCKS_PER_CU;
    if (s->max_blocks) cap = std::min<uint64_t>(cap, s->max_blocks);
    const uint32_t nb = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(tiles, cap));
    const uint32_t mask = (1u << bits) - 1;
    uint32_t* table = (uint32_t*) s->table.ptr;
    uint32_t* totals = table + (size_t) RADIX * nb;

    using Smem = ScatterSmem<KeyT, BITS, G::THREADS, G::KPT, G::CARRY, 1, VALS>;
    auto scatter = radix_scatter_kernel<KeyT, BITS, G::THREADS, G::KPT, G::CARRY, 0, false, G::BLOCKS_PER_CU * G::THREADS / 256, 1, false, false, XF, VALS>;
    // launch-bound sizes (small geometry, few workgroups): the scatter sums the counts itself, no row-scan launch
    auto scatter_fused = radix_scatter_kernel<KeyT, BITS, G::THREADS, G::KPT, G::CARRY, 0, false, G::BLOCKS_PER_CU * G::THREADS / 256, 1, false, false, XF, VALS, !LARGE>;
    // (THREADS / RADIX threads share a digit's row in that prologue.  8-bit digits, 2 sharers: up to 32 workgroups -- with 64 the
    // prologue costs more than the launch, 2^18 pairs 54 -> 60 us.  4-bit digits, 32 sharers: up to 256 workgroups, 2^18 pairs 88
    // -> 75 us, 2^20: 114 -> 106 us; with 512 every workgroup reads 32 KiB of table and the pass loses again.)
    constexpr uint32_t kRowSharers = G::THREADS / RADIX >= 1 ? G::THREADS / RADIX : 1;
    constexpr uint32_t kFusedLimit = kFusedScanMaxBlocks * (kRowSharers / 4 >= 1 ? kRowSharers / 4 : 1);
    const bool fused = !LARGE && nb <= kFusedLimit && !histogram_out && !s->no_fused_scan && !pa.plan;
    static std::once_flag lds_opt_in; // per instantiation: allow > 64 KiB of dynamic LDS (handles may live on several threads)
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
        if (lds_opt_in_result == hipSuccess)
            lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);

    s->mark(stream);
    // the count kernel only shares TILE and the grid with the scatter kernel; 1024 threads keep enough loads in flight
    // when there is one workgroup per CU
    constexpr int COUNT_THREADS = LARGE ? 1024 : G::THREADS;
    // (the first pass of a planned sort of unsigned keys runs the instantiation that also notes which key bits vary)
    auto count_plain = radix_count_kernel<KeyT, BITS, COUNT_THREADS, G::TILE, XF, false>;
    auto count_collect = radix_count_kernel<KeyT, BITS, COUNT_THREADS, G::TILE, XF, !XF>;
    hipLaunchKernelGGL((pa.flags & kPlanCollectBits) ? count_collect : count_plain, dim3(nb), dim3(COUNT_THREADS), 0, stream,
                       src_k, table, (uint32_t) count, shift, mask, tiles, xform, (const KeyT*) dst_k, pa.plan, pa.pass, false,
                       pa.flags, 0u);
    HIP_TRY(hipGetLastError()); // every launch is checked where it happens: a failed count launch is reported as such
    s->mark(stream);
    if (!fused)
    {
        hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, stream, table, totals, nb, (uint32_t) count,
                           pa.may_skip ? pa.plan : (PassPlan*) nullptr, pa.pass, 0u, pa.plan, (uint32_t*) nullptr, 1u);
        HIP_TRY(hipGetLastError());
    }
    s->mark(stream, true);
    if (histogram_out)
    {
        HIP_TRY(hipMemcpyAsync(histogram_out, totals, ((size_t) 1 << bits) * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                               stream));
        if (s->after_histogram_event) HIP_TRY(hipEventRecord(s->after_histogram_event, stream));
    }
    hipLaunchKernelGGL(fused ? scatter_fused : scatter, dim3(nb), dim3(G::THREADS), sizeof(Smem), stream, src_k, src_v, dst_k,
                       dst_v, (const uint32_t*) table, (const uint32_t*) totals, (uint32_t) count, shift, mask, tiles,
                       (unsigned long long*) nullptr, xform, pa.plan, pa.pass, (uint32_t*) nullptr);
    s->mark(stream, true);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

// Large inputs of 4-byte keys in 16-byte aligned arrays: the same three launches with the 128-byte-line scatter kernel
// (its own tile size, shared by the count kernel: both cut the input with block_tile_range).
template<typename KeyT, int BITS, bool XF, bool VALS>
glu_status launch_pass_lines(glu_radix_sort_s* s, const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v,
                             size_t count, uint32_t shift, uint32_t bits, uint32_t* histogram_out, hipStream_t stream,
                             uint32_t xform, PlanArgs pa)
{
    using G = LinesGeometry<KeyT, BITS, VALS>;
    constexpr int RADIX = 1 << BITS;
    const uint32_t tiles = (uint32_t) ((count + G::TILE - 1) / G::TILE);
    uint64_t cap = (uint64_t) usable_cus(s);
    if (s->max_blocks) cap = std::min<uint64_t>(cap, s->max_blocks);
    const uint32_t nb = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(tiles, cap));
    const uint32_t mask = (1u << bits) - 1;
    uint32_t* table = (uint32_t*) s->table.ptr;
    uint32_t* totals = table + (size_t) RADIX * nb;

    // Whole tiles per workgroup (share = 0).  GLU_HIP_SORT_EQUAL_SHARES=1 (tuning) gives every workgroup the same number of
    // elements instead (a multiple of 64): it takes the sawtooth out of sort time over n (with 2.3 tiles per workgroup a
    // whole-tile split has some do 3 and the rest wait: 6 M pairs 237 -> 199 us, 8 M: 283 -> 241 us), but then every
    // workgroup ends on a partial tile, whose guarded, un-prefetched loads cost as much as a whole tile: 5 M 174 -> 188 us,
    // 13 M 290 -> 327 us, 2^28 level.  Not the default.
    const uint32_t share = s->equal_shares ? (uint32_t) ((((uint64_t) count + nb - 1) / nb + 63u) & ~(uint64_t) 63u) : 0u;
    using Smem = LineSmem<KeyT, BITS, G::THREADS, G::KPT, VALS>;
    // the line stores are non-temporal: what a pass writes is next read by the count kernel of the following pass, a
    // once-through stream that runs at full speed only if the lines are not sitting dirty in L2 / Infinity Cache
    // (count behind a plain-store scatter 0.206 ms, behind a non-temporal one 0.160 ms; the scatter itself is level)
    constexpr int RS = (G::KPT + 2) / 3;
    auto scatter_nt = radix_scatter_lines_kernel<KeyT, BITS, G::THREADS, G::KPT, XF, VALS, 0, false, RS, true, true>;
    auto scatter_plain = radix_scatter_lines_kernel<KeyT, BITS, G::THREADS, G::KPT, XF, VALS, 0, false, RS, true, false>;
    // the same kernel under another name for the ordinary passes enqueued behind an attempt to end in LDS: they return at
    // once when the attempt was accepted, and a kernel trace's per-name statistics of the scatter stay those of launches
    // that moved data
    auto scatter_behind = scatter_nt;
    constexpr bool kHasBehind = BITS == 8 && !XF; // (the sorts that make such attempts)
    if constexpr (kHasBehind)
        scatter_behind = radix_scatter_lines_kernel<KeyT, BITS, G::THREADS, G::KPT, XF, VALS, 0, false, RS, true, true, 0, false, false, true>;
    static std::once_flag lds_opt_in; // per instantiation: allow > 64 KiB of dynamic LDS
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_nt, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
        if (lds_opt_in_result == hipSuccess)
            lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
        if (lds_opt_in_result == hipSuccess && kHasBehind)
            lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_behind, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);
    // ... when the arrays are far larger than the 256 MiB Infinity Cache.  Below about 320 MiB of keys + values plain stores
    // win: part of what a pass wrote is still cached when the next pass reads it (4-byte keys with values, 4.3 M .. 36 M pairs:
    // 3-7 % of the sort; 44 M and up: non-temporal ahead; profiles/r05/nt_stores_by_size.txt)
    const bool nt = s->nt_stores && (uint64_t) count * (sizeof(KeyT) + (VALS ? sizeof(uint32_t) : 0)) >= s->nt_min_bytes;
    auto scatter = nt ? scatter_nt : scatter_plain;
    if (pa.behind_attempt && kHasBehind && nt) scatter = scatter_behind;

    const uint2* ranges = nullptr;
    s->mark(stream);
    uint32_t* const sub_table = (uint32_t*) s->pair_sub.ptr; // 4-bit digits: the leader's table per sub-block
    if constexpr (!XF || BITS == 8)
    {
        if (pa.pair_role == 1)
        {
            // leader: one read of the keys for this pass's table and the two-digit table the follower's comes from
            if constexpr (BITS == 8)
            {
                // (XF: the first top-bit pass of a typed sort that tries to end in LDS encodes on load; it never collects key bits)
                auto count2_plain = radix_pair_count_kernel<KeyT, G::TILE, XF, false>;
                auto count2_collect = radix_pair_count_kernel<KeyT, G::TILE, XF, !XF>; // first pass: also notes which key bits vary
                static std::once_flag count2_opt_in;
                static hipError_t count2_opt_in_result = hipSuccess;
                std::call_once(count2_opt_in, [&] {
                    count2_opt_in_result = hipFuncSetAttribute((const void*) count2_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(PairCountSmem));
                    if (count2_opt_in_result == hipSuccess)
                        count2_opt_in_result = hipFuncSetAttribute((const void*) count2_collect, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(PairCountSmem));
                });
                HIP_TRY(count2_opt_in_result);
                auto count2 = (pa.flags & kPlanCollectBits) ? count2_collect : count2_plain;
                // (the leader of an attempt to end in LDS puts wrapped 16-bit counters right: exact run lengths whatever the keys)
                hipLaunchKernelGGL(count2, dim3(nb), dim3(1024), sizeof(PairCountSmem), stream, src_k, table, (uint32_t*) s->pair_t2.ptr,
                                   (uint32_t) count, shift, mask, pa.shift2, (1u << pa.bits2) - 1, tiles, xform, (const KeyT*) dst_k,
                                   pa.plan, pa.pass, pa.flags, share, pa.finish_geo_first ? (uint32_t*) s->pair_wide.ptr : (uint32_t*) nullptr);
            }
            else
            {
                auto count4_plain = radix_pair4_count_kernel<KeyT, G::TILE, false, false>;
                auto count4_collect = radix_pair4_count_kernel<KeyT, G::TILE, false, true>;
                hipLaunchKernelGGL((pa.flags & kPlanCollectBits) ? count4_collect : count4_plain,
                                   dim3(nb * kPairSub), dim3(256), 0, stream, src_k, sub_table, (uint32_t*) s->pair_t2.ptr, (uint32_t) count, shift, mask, pa.shift2,
                                   (1u << pa.bits2) - 1, tiles, xform, (const KeyT*) dst_k, pa.plan, pa.pass, pa.flags, share);
            }
            HIP_TRY(hipGetLastError());
        }
    }
    if (pa.pair_role == 2)
    {
        // follower: unit runs and their digit counts from the leader's tables (s->table / pair_sub still hold them); its own
        // table lives in pair_table.  The count kernel behind it runs only if a kernel before it asked for that.
        uint32_t* leader_table = table;
        table = (uint32_t*) s->pair_table.ptr;
        totals = table + (size_t) RADIX * nb;
        ranges = (const uint2*) s->pair_ranges.ptr;
        if constexpr (BITS == 8)
        {
            hipLaunchKernelGGL(radix_pair_unitsum_kernel, dim3(nb), dim3(1024), 0, pa.unitsum_side ? s->side : stream, (const uint32_t*) s->pair_t2.ptr,
                               (const uint32_t*) leader_table, (const uint32_t*) (leader_table + (size_t) RADIX * nb), table,
                               (uint2*) s->pair_ranges.ptr, (uint32_t) count, pa.plan, pa.pass, shift, mask, pa.flags);
            if (pa.unitsum_side)
            {
                HIP_TRY(hipEventRecord(s->ev_unit, s->side));
                HIP_TRY(hipStreamWaitEvent(stream, s->ev_unit, 0));
            }
        }
        else
            hipLaunchKernelGGL(radix_pair4_unitsum_kernel, dim3(nb), dim3(1024), 0, stream, (const uint32_t*) s->pair_t2.ptr,
                               (const uint32_t*) sub_table, (const uint32_t*) (leader_table + (size_t) RADIX * nb), table,
                               (uint2*) s->pair_ranges.ptr, (uint32_t) count, pa.plan, pa.pass, shift, mask, pa.flags);
        HIP_TRY(hipGetLastError());
    }
    if (pa.pair_role != 1)
    {
        auto count_plain = radix_count_kernel<KeyT, BITS, 1024, G::TILE, XF, false>;
        auto count_collect = radix_count_kernel<KeyT, BITS, 1024, G::TILE, XF, !XF>;
        hipLaunchKernelGGL((pa.flags & kPlanCollectBits) ? count_collect : count_plain, dim3(nb), dim3(1024), 0, stream, src_k, table,
                           (uint32_t) count, shift, mask, tiles, xform, (const KeyT*) dst_k, pa.plan, pa.pass, pa.pair_role == 2,
                           pa.flags, share);
        HIP_TRY(hipGetLastError());
    }
    s->mark(stream);
    if (BITS == 4 && pa.pair_role == 1)
    {
        // 4-bit leader: its count kernel wrote the table per sub-block; one scan gives where every (digit value, sub-block)
        // unit starts, the digit totals, and -- every kPairSub-th entry -- the usual per-block table of the scatter
        hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, stream, sub_table, totals, nb * kPairSub,
                           (uint32_t) count, pa.may_skip ? pa.plan : (PassPlan*) nullptr, pa.pass, 0u, pa.plan, table,
                           kPairSub);
    }
    else
    {
        // (8-bit leader: a unit longer than 1/16 of a workgroup's share makes the follower count for itself; the units of a
        // 4-bit leader are parts of sub-blocks, which are that short by construction)
        hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, stream, table, totals, nb, (uint32_t) count,
                           pa.may_skip ? pa.plan : (PassPlan*) nullptr, pa.pass,
                           pa.pair_role == 1 && BITS == 8
                               ? (s->pair_unit_div ? std::max<uint32_t>(1u, (uint32_t) (count / nb / s->pair_unit_div)) : 0xFFFFFFFFu)
                               : 0u,
                           pa.plan);
    }
    HIP_TRY(hipGetLastError());
    if (pa.finish_geo_first)
    {
        hipLaunchKernelGGL(radix_finish_lengths_kernel, dim3(kPairRadix), dim3(1024), 0, stream, (const uint32_t*) s->pair_t2.ptr, nb,
                           (uint32_t*) s->finish_lengths.ptr, (const PassPlan*) pa.plan, pa.pass, (const uint32_t*) s->pair_wide.ptr);
        HIP_TRY(hipGetLastError());
        const bool long_runs = pa.finish_long_ok && s->long_image.ptr;
        hipLaunchKernelGGL(radix_finish_plan_kernel, dim3(kFinishPlanBlocks), dim3(1024), 0, stream, (const uint32_t*) s->finish_lengths.ptr,
                           (uint32_t*) s->finish_starts.ptr, (uint32_t) count, pa.finish_geo_first, pa.finish_geo_last, pa.plan, pa.pass,
                           pa.finish_first_ordinary, pa.finish_num_ordinary, s->finish_hint, pa.finish_seq, pa.finish_top_bit,
                           pa.finish_key_bits, long_runs ? 1u : 0u, (uint32_t*) s->finish_crowded.ptr, (uint32_t*) s->finish_outcomes.ptr,
                           (uint32_t) (sizeof(KeyT) + (VALS ? sizeof(uint32_t) : 0)));
        HIP_TRY(hipGetLastError());
        if (long_runs)
        {
            // (on the caller's queue: on the side stream it finds no room beside the scatter's 1024-thread workgroups, finishes when that
            // does, and holds up the follower's unit sums behind it -- measured: + 17 us)
            hipLaunchKernelGGL(radix_finish_long_runs_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*) s->finish_starts.ptr,
                               (const PassPlan*) pa.plan, usable_cus(s) * kLongRunsSharesPerWg, (uint32_t*) s->long_image.ptr, (uint32_t*) s->long_hdr.ptr);
            HIP_TRY(hipGetLastError());
        }
        if (pa.fork_side) // the decision is made: the sequence that is expected not to run leaves the caller's queue here
        {
            HIP_TRY(hipEventRecord(s->ev_fork, stream));
            HIP_TRY(hipStreamWaitEvent(s->side, s->ev_fork, 0));
        }
    }
    s->mark(stream, true);
    if (histogram_out)
    {
        HIP_TRY(hipMemcpyAsync(histogram_out, totals, ((size_t) 1 << bits) * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                               stream));
        if (s->after_histogram_event) HIP_TRY(hipEventRecord(s->after_histogram_event, stream));
    }
    hipLaunchKernelGGL(scatter, dim3(nb), dim3(G::THREADS), sizeof(Smem), stream, src_k, src_v, dst_k, dst_v,
                       (const uint32_t*) table, (const uint32_t*) totals, (uint32_t) count, shift, mask, tiles,
                       (unsigned long long*) nullptr, xform, pa.plan, pa.pass, ranges, share, (const uint32_t*) nullptr, (const uint32_t*) nullptr, 0u, 0u, (const uint32_t*) nullptr);
    s->mark(stream, true);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

template<typename KeyT, int BITS>
glu_status launch_pass_sized(glu_radix_sort_s* s, const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v,
                             size_t count, uint32_t shift, uint32_t bits, uint32_t* histogram_out, hipStream_t stream,
                             uint32_t xform = 0, PlanArgs pa = PlanArgs())
{
    // Keys-only sorts (no value arrays) run the VALS = false instantiations.
    const bool vals = src_v != nullptr;
    const size_t large_tile = vals ? GeometryFor<KeyT, BITS, true, true>::TILE : GeometryFor<KeyT, BITS, true, false>::TILE;
    const bool large = count >= large_tiles_from<KeyT, BITS>(s, vals, large_tile) && !s->force_small;
    {
        if (lines_applicable<KeyT, BITS>(s, src_k, src_v, dst_k, dst_v, count))
        {
#define GLU_LAUNCH_LINES(XF_, VALS_) \
    launch_pass_lines<KeyT, BITS, XF_, VALS_>(s, src_k, src_v, dst_k, dst_v, count, shift, bits, histogram_out, stream, XF_ ? xform : 0u, pa)
            if (vals) return xform ? GLU_LAUNCH_LINES(true, true) : GLU_LAUNCH_LINES(false, true);
            return xform ? GLU_LAUNCH_LINES(true, false) : GLU_LAUNCH_LINES(false, false);
#undef GLU_LAUNCH_LINES
        }
    }
#define GLU_LAUNCH(LARGE_, XF_, VALS_) \
    launch_pass<KeyT, BITS, LARGE_, XF_, VALS_>(s, src_k, src_v, dst_k, dst_v, count, shift, bits, histogram_out, stream, XF_ ? xform : 0u, pa)
    if (vals)
    {
        if (xform) return large ? GLU_LAUNCH(true, true, true) : GLU_LAUNCH(false, true, true);
        return large ? GLU_LAUNCH(true, false, true) : GLU_LAUNCH(false, false, true);
    }
    if (xform) return large ? GLU_LAUNCH(true, true, false) : GLU_LAUNCH(false, true, false);
    return large ? GLU_LAUNCH(true, false, false) : GLU_LAUNCH(false, false, false);
#undef GLU_LAUNCH
}

template<typename KeyT>
glu_status dispatch_pass(glu_radix_sort_s* s, const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v,
                         size_t count, uint32_t shift, uint32_t bits, uint32_t* histogram_out, hipStream_t stream,
                         uint32_t xform, PlanArgs pa)
{
    if (bits <= 4) return launch_pass_sized<KeyT, 4>(s, src_k, src_v, dst_k, dst_v, count, shift, bits, histogram_out, stream, xform, pa);
    return launch_pass_sized<KeyT, 8>(s, src_k, src_v, dst_k, dst_v, count, shift, bits, histogram_out, stream, xform, pa);
}
} // namespace host
} // namespace glu_hip
