// glu_sort_finish.hip -- the launches of the in-LDS pass of a sort that ends in LDS (glu_sort_launch.hpp; kernels:
// radix_lds_bucket.hpp, radix_lds_finish.hpp).
#include <algorithm>
#include <mutex>

#include "glu_sort_launch.hpp"
#include "radix_lds_bucket.hpp"

namespace glu_hip
{
namespace host
{
template<int THREADS, int KPT>
glu_status launch_seg_finish_geo(const uint32_t* src_k, const uint32_t* src_v, uint32_t* dst_k, uint32_t* dst_v, const uint32_t* starts,
                                 uint32_t nruns, uint32_t geo, uint32_t split_log2, uint32_t low_bits, const uint32_t* gate, uint32_t gate_cap,
                                 uint32_t rank_bits, hipStream_t stream)
{
    using Smem = FinishSmem<uint32_t, THREADS, KPT, true>;
    auto sort_kernel = radix_finish_sort_kernel<uint32_t, THREADS, KPT, true, false, false>;
    auto ranges_kernel = radix_finish_ranges_kernel<uint32_t, THREADS, KPT, true, false>;
    static std::once_flag lds_opt_in;
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
        if (lds_opt_in_result == hipSuccess)
            lds_opt_in_result = hipFuncSetAttribute((const void*) ranges_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);
    const uint32_t rank_from = finish_rank_from(low_bits, rank_bits);
    if (split_log2 == 0)
    {
        hipLaunchKernelGGL(sort_kernel, dim3(nruns), dim3(THREADS), sizeof(Smem), stream, const_cast<uint32_t*>(src_k), const_cast<uint32_t*>(src_v),
                           dst_k, dst_v, starts, low_bits, (const PassPlan*) nullptr, 0u, geo, 0u, nruns, gate, gate_cap, rank_bits,
                           (unsigned long long*) nullptr, (const uint32_t*) nullptr, 0u);
        HIP_TRY(hipGetLastError());
    }
    const uint64_t items = (((uint64_t) nruns + 7u) & ~7ull) << split_log2;
    // (the long runs alone: workgroups that loop over the runs and skip the short ones)
    const uint32_t grid = (uint32_t) (split_log2 == 0 ? std::min<uint64_t>(items, 2048u) : items);
    hipLaunchKernelGGL(ranges_kernel, dim3(grid), dim3(THREADS), sizeof(Smem), stream, src_k, src_v, dst_k, dst_v, starts, nruns, low_bits,
                       rank_from, split_log2, split_log2 == 0 ? (uint32_t) (THREADS * KPT) : 0u, gate, gate_cap, 0u);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

glu_status launch_seg_finish(const uint32_t* src_k, const uint32_t* src_v, uint32_t* dst_k, uint32_t* dst_v, const uint32_t* starts,
                                    uint32_t nruns, uint32_t geo, uint32_t split_log2, uint32_t low_bits, const uint32_t* gate, uint32_t gate_cap,
                                    uint32_t rank_bits, hipStream_t stream)
{
#define GLU_SEG_FINISH(GEO_, THREADS_, KPT_)                                                                                                 \
    if (geo == GEO_)                                                                                                                         \
    {                                                                                                                                        \
        static_assert(finish_geometry_capacity(GEO_) == THREADS_ * KPT_, "geometry table");                                                  \
        return launch_seg_finish_geo<THREADS_, KPT_>(src_k, src_v, dst_k, dst_v, starts, nruns, geo, split_log2, low_bits, gate, gate_cap, rank_bits, stream); \
    }
    GLU_SEG_FINISH(1, 256, 6)
    GLU_SEG_FINISH(2, 256, 10)
    GLU_SEG_FINISH(3, 256, 18)
    GLU_SEG_FINISH(4, 512, 18)
    GLU_SEG_FINISH(5, 1024, 17)
#undef GLU_SEG_FINISH
    return fail(GLU_ERROR_INVALID_STATE, "no such tile geometry: %u", geo);
}

template<typename KeyT, bool VALS, bool XF>
glu_status launch_finish(KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b, uint32_t* vals_b, const uint32_t* starts,
                         uint32_t geo_first, uint32_t geo_last, uint32_t geo_expected, uint32_t low_bits, const PassPlan* plan,
                         uint32_t pass, uint32_t key_xf, hipStream_t stream, uint32_t rank_bits, uint32_t* crowded,
                         FinishMarks marks)
{
    // (marks: the two profile marks of the in-LDS pass go around the launch of the EXPECTED tile, the one that does the work in
    // a timed loop -- not around the launches beside it that return at once)
    constexpr uint32_t nruns = kFinishRuns;
    // (order: the geometries that are not expected first -- they return at once in front of the long kernel instead of waiting
    // behind it for room on the CUs; what follows the expected one is the launch that takes its crowded runs)
#define GLU_FINISH(GEO_, THREADS_, KPT_)                                                                                          \
    if (geo_first <= GEO_ && GEO_ <= geo_last && (GEO_ == geo_expected) == expected_turn)                                         \
    {                                                                                                                             \
        static_assert(finish_geometry_capacity(GEO_) == THREADS_ * KPT_, "geometry table");                                       \
        using Smem = BucketSmem<KeyT, THREADS_, KPT_, VALS>;                                                                      \
        auto kern = GEO_ == geo_expected ? radix_finish_bucket_kernel<KeyT, THREADS_, KPT_, VALS, false, XF>                      \
                                         : radix_finish_bucket_kernel<KeyT, THREADS_, KPT_, VALS, true, XF>;                      \
        static std::once_flag lds_opt_in;                                                                                         \
        static hipError_t lds_opt_in_result = hipSuccess;                                                                         \
        std::call_once(lds_opt_in, [&] {                                                                                          \
            for (const void* k : {(const void*) radix_finish_bucket_kernel<KeyT, THREADS_, KPT_, VALS, false, XF>,                \
                                  (const void*) radix_finish_bucket_kernel<KeyT, THREADS_, KPT_, VALS, true, XF>})                \
                if (lds_opt_in_result == hipSuccess)                                                                              \
                    lds_opt_in_result = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));   \
        });                                                                                                                       \
        HIP_TRY(lds_opt_in_result);                                                                                               \
        if (GEO_ == geo_expected) marks(stream);                                                                                    \
        hipLaunchKernelGGL(kern, dim3(GEO_ == geo_expected ? nruns : std::min(nruns, 8192u)), dim3(THREADS_), sizeof(Smem),       \
                           stream, keys_a, vals_a, keys_b, vals_b, starts, low_bits, plan, pass, (uint32_t) GEO_, key_xf, nruns,  \
                           crowded);                                                                                              \
        if (GEO_ == geo_expected) marks(stream);                                                                                    \
        HIP_TRY(hipGetLastError());                                                                                               \
    }
    // The runs the bucket kernel listed (or all of them: PassPlan::finish_rounds) by ballot rounds, 8192 workgroups that loop, split
    // by the runs' length over two launches: the tile the sort is expected to take (or the one below the largest, if that is the
    // expected one) takes the runs that fit it -- four workgroups per CU for 256 x 18: 24-bit keys 3.15 -> 2.95 ms at 2^28 against
    // one launch in the largest tile --, the largest enqueued tile takes the longer ones (runs longer than the tile the device
    // CHOSE are the segmented passes' either way).
#define GLU_FINISH_ROUNDS(GEO_, THREADS_, KPT_, LEN_ABOVE_)                                                                       \
    {                                                                                                                             \
        using Smem = FinishSmem<KeyT, THREADS_, KPT_, VALS>;                                                                      \
        auto kern = radix_finish_sort_kernel<KeyT, THREADS_, KPT_, VALS, true, XF>;                                               \
        static std::once_flag lds_opt_in;                                                                                         \
        static hipError_t lds_opt_in_result = hipSuccess;                                                                         \
        std::call_once(lds_opt_in, [&] {                                                                                          \
            lds_opt_in_result = hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)); \
        });                                                                                                                       \
        HIP_TRY(lds_opt_in_result);                                                                                               \
        hipLaunchKernelGGL(kern, dim3(8192), dim3(THREADS_), sizeof(Smem), stream, keys_a, vals_a, keys_b, vals_b, starts,        \
                           low_bits, plan, pass, 0u, key_xf, nruns, (const uint32_t*) nullptr, 0u, rank_bits,                     \
                           (unsigned long long*) nullptr, (const uint32_t*) crowded, (uint32_t) (LEN_ABOVE_));                    \
        HIP_TRY(hipGetLastError());                                                                                               \
    }
    for (int turn = 0; turn < 2; turn++)
    {
        const bool expected_turn = turn == 1;
        GLU_FINISH(1, 256, 6)
        GLU_FINISH(2, 256, 10)
        if constexpr (sizeof(KeyT) == 4)
        {
            GLU_FINISH(3, 256, 18)
            GLU_FINISH(4, 512, 18)
        }
        else
        {
            // 8-byte keys: twice the waves per workgroup (the word stage is 8 bytes per slot: two workgroups per CU)
            GLU_FINISH(3, 512, 9)
            GLU_FINISH(4, 1024, 9)
        }
    }
    // (the tile for the shorter runs, then -- if more than one tile is enqueued -- the largest for the runs that one leaves)
    const uint32_t geo_lo = geo_expected < geo_last ? geo_expected : std::max(geo_first, geo_last - 1u);
    for (uint32_t turn = 0; turn < (geo_lo < geo_last ? 2u : 1u); turn++)
    {
        const uint32_t g = turn == 0 ? geo_lo : geo_last;
        const uint32_t len_above = turn == 0 ? 0u : finish_geometry_capacity(geo_lo);
        if (g == 1) GLU_FINISH_ROUNDS(1, 256, 6, len_above)
        if (g == 2) GLU_FINISH_ROUNDS(2, 256, 10, len_above)
        if constexpr (sizeof(KeyT) == 4)
        {
            if (g == 3) GLU_FINISH_ROUNDS(3, 256, 18, len_above)
            if (g == 4) GLU_FINISH_ROUNDS(4, 512, 18, len_above)
        }
        else
        {
            if (g == 3) GLU_FINISH_ROUNDS(3, 512, 9, len_above)
            if (g == 4) GLU_FINISH_ROUNDS(4, 1024, 9, len_above)
        }
    }
#undef GLU_FINISH
#undef GLU_FINISH_ROUNDS
    return GLU_OK;
}

#define GLU_LAUNCH_FINISH_DEFINE(K_, V_, X_)                                                                                      \
    template glu_status launch_finish<K_, V_, X_>(K_*, uint32_t*, K_*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t,  \
                                                  uint32_t, const PassPlan*, uint32_t, uint32_t, hipStream_t, uint32_t,          \
                                                  uint32_t*, FinishMarks);
GLU_LAUNCH_FINISH_INSTANCES(GLU_LAUNCH_FINISH_DEFINE)
#undef GLU_LAUNCH_FINISH_DEFINE
} // namespace host
} // namespace glu_hip
