// glu_sort_launch.hpp -- the launches of the in-LDS pass (defined in glu_sort_finish.hip, called from glu_hip.hip): a translation
// unit of their own because radix_finish_bucket_kernel and radix_finish_sort_kernel are a hundred instantiations.
#pragma once
#include <cmath>

#include "glu_host.hpp"
#include "radix_lds_finish.hpp"

namespace glu_hip
{
namespace host
{
// A sort that ends in LDS (radix_lds_finish.hpp): the tile geometry of its last pass that suits uniformly drawn keys -- the
// longest of 65536 runs stays below mean + 6 sigma.  The launches of that geometry and of the next larger ones are enqueued
// and the device picks by the longest run it counted.  0: no geometry holds such runs (more than about 2^29 pairs).
inline uint32_t finish_geometry_for(size_t count, size_t runs = kFinishRuns, uint32_t geometries = kFinishGeometries)
{
    const double mean = (double) count / (double) runs;
    const double need = mean + 6.0 * std::sqrt(mean) + 8.0;
    for (uint32_t g = 1; g <= geometries; g++)
        if (need <= (double) finish_geometry_capacity(g)) return g;
    return 0;
}

// More than 16 bits left to order (64-bit keys): the rounds of the in-LDS pass rank the top 16 .. 23 of them and ties are repaired
// exactly (radix_lds_finish.hpp; 4-byte keys run every round).  glu_radix_sort_s::finish_rank_bits.
inline uint32_t finish_rank_from(uint32_t low_bits, uint32_t rank_bits)
{
    return low_bits > rank_bits ? ((low_bits - rank_bits) / 8u) * 8u : 0u;
}

// The two profile marks of the in-LDS pass go around the launch of the EXPECTED tile, the one that does the work in a timed
// loop -- not around the launches beside it that return at once (glu_radix_sort_s::mark, through this pair: the sort object's
// type stays in glu_hip.hip).
struct FinishMarks
{
    void (*mark)(void* object, hipStream_t stream) = nullptr;
    void* object = nullptr;
    void operator()(hipStream_t stream) const
    {
        if (mark) mark(object, stream);
    }
};

// The in-LDS pass of a segmented sort, src -> dst over `nruns` runs (seg_run_plan): tile geometry `geo` (1 .. 4); split_log2 = 0: a
// workgroup per run for the runs that fit the tile and radix_finish_ranges_kernel behind it for the longer ones; split_log2 > 0:
// every run is split over 2^split_log2 workgroups of the ranges kernel.  Both return at once if the longest run (*gate) is
// beyond gate_cap.
glu_status launch_seg_finish(const uint32_t* src_k, const uint32_t* src_v, uint32_t* dst_k, uint32_t* dst_v, const uint32_t* starts,
                             uint32_t nruns, uint32_t geo, uint32_t split_log2, uint32_t low_bits, const uint32_t* gate, uint32_t gate_cap,
                             uint32_t rank_bits, hipStream_t stream);

// The in-LDS pass of a whole-key sort (round 6): radix_finish_bucket_kernel for every enqueued tile geometry -- the one the sort is
// expected to take gets a workgroup per run, the others 8192 workgroups that loop --, and behind them round 5's ballot-ranked
// kernel for the runs the bucket kernel listed as crowded (or for all of them: PassPlan::finish_rounds), two launches split by the
// runs' length: the expected tile for the runs that fit it, the largest enqueued tile for the longer ones.  They return at once
// when the lists are empty.
template<typename KeyT, bool VALS, bool XF>
glu_status launch_finish(KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b, uint32_t* vals_b, const uint32_t* starts,
                         uint32_t geo_first, uint32_t geo_last, uint32_t geo_expected, uint32_t low_bits, const PassPlan* plan,
                         uint32_t pass, uint32_t key_xf, hipStream_t stream, uint32_t rank_bits, uint32_t* crowded,
                         FinishMarks marks);
#define GLU_LAUNCH_FINISH_INSTANCES(X_)                                                                                           \
    X_(uint32_t, true, false) X_(uint32_t, true, true) X_(uint32_t, false, false) X_(uint32_t, false, true)                       \
    X_(uint64_t, true, false) X_(uint64_t, true, true) X_(uint64_t, false, false) X_(uint64_t, false, true)
#define GLU_LAUNCH_FINISH_EXTERN(K_, V_, X_)                                                                                      \
    extern template glu_status launch_finish<K_, V_, X_>(K_*, uint32_t*, K_*, uint32_t*, const uint32_t*, uint32_t, uint32_t,     \
                                                         uint32_t, uint32_t, const PassPlan*, uint32_t, uint32_t, hipStream_t,   \
                                                         uint32_t, uint32_t*, FinishMarks);
GLU_LAUNCH_FINISH_INSTANCES(GLU_LAUNCH_FINISH_EXTERN)
#undef GLU_LAUNCH_FINISH_EXTERN
} // namespace host
} // namespace glu_hip
