// glu_sort_passes_u64.hip -- the counting-pass launchers of libglu_hip.so for 8-byte keys (glu_sort_passes.hpp).
#include "glu_sort_passes.hpp"

namespace glu_hip
{
namespace host
{
template glu_status dispatch_pass<uint64_t>(glu_radix_sort_s*, const uint64_t*, const uint32_t*, uint64_t*, uint32_t*, size_t, uint32_t, uint32_t, uint32_t*,
                                      hipStream_t, uint32_t, PlanArgs);
} // namespace host
} // namespace glu_hip
