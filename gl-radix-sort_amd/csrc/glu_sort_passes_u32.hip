// glu_sort_passes_u32.hip -- the counting-pass launchers of libglu_hip.so for 4-byte keys (glu_sort_passes.hpp).
#include "glu_sort_passes.hpp"

namespace glu_hip
{
namespace host
{
template glu_status dispatch_pass<uint32_t>(glu_radix_sort_s*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, size_t, uint32_t, uint32_t, uint32_t*,
                                      hipStream_t, uint32_t, PlanArgs);
} // namespace host
} // namespace glu_hip
