// glu_host.hpp -- what the translation units of libglu_hip.so share on the HOST side: the error slot behind glu_last_error(), the
// one device of the process and its queue, the buffer registry behind the glu_buffer handles, grow-only scratch allocations.
// Defined in glu_core.hip.  The library is built from six translation units (round 6; it was one, a three-minute compile):
//   glu_core.hip              errors, device, buffers, timer
//   glu_hip.hip               RadixSort's launch sequences, the segmented sort and the sharded sort (glu_dist_impl.hpp)
//   glu_sort_passes_u32.hip   the launchers of one counting pass for 4-byte keys (glu_sort_passes.hpp over glu_sort_object.hpp)
//   glu_sort_passes_u64.hip   the same for 8-byte keys
//   glu_sort_finish.hip       the launches of the in-LDS pass (radix_lds_bucket.hpp, radix_lds_finish.hpp: a hundred kernel instantiations)
//   glu_scan_reduce.hip       BlellochScan and Reduce
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

#define GLU_HIP_BUILD 1
#include "glu_hip.h"

namespace glu_hip
{
namespace host
{
extern thread_local std::string g_last_error;

// GLU_VERBOSE=1: allocations and placement searches are narrated on stderr (read once per process)
bool glu_verbose();

// Every other environment variable the library reads goes through here: defaults of new sort objects (kSortOptions), the
// tuning lists of the placement search / scan / reduce, and the test hooks of glu_dist (fault injection, the RCCL test double).
const char* glu_env(const char* name);

glu_status fail(glu_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? GLU_ERROR_OUT_OF_MEMORY : GLU_ERROR_DEVICE, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                                        \
    } while (0)

#define GLU_TRY(expr)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        glu_status s_ = (expr);                                                                                        \
        if (s_ != GLU_OK) return s_;                                                                                   \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// device / queue state (one device per process)
// ------------------------------------------------------------------------------------------------------------
struct Device
{
    std::mutex mutex;
    bool ready = false;
    int requested = -1;
    int id = 0;
    int num_cus = 256;
    hipStream_t queue = nullptr;
    hipDeviceProp_t props;
};
extern Device g_dev;

glu_status ensure_device();

// Every public entry point runs this first: the device is initialised once per process, and the calling thread's
// current HIP device is made the library's device (a new thread's current device is 0, and an application may switch
// devices between two library calls -- torch.cuda.device(k), hipSetDevice -- so the current device is asked for every
// time instead of being remembered per thread: without this, scratch would be allocated on one device and the kernels
// launched on a stream of another).
glu_status enter();

inline hipStream_t pick_stream(void* stream) { return stream ? (hipStream_t) stream : g_dev.queue; }

// ------------------------------------------------------------------------------------------------------------
// buffers
// ------------------------------------------------------------------------------------------------------------
struct Buffer
{
    void* ptr = nullptr;
    size_t size = 0;
    bool owned = true;
};
glu_status lookup(glu_buffer h, Buffer& out, const char* what);

// grow-only device allocation owned by an operator object
struct Scratch
{
    void* ptr = nullptr;
    size_t size = 0;
    glu_status reserve(size_t bytes)
    {
        if (bytes <= size) return GLU_OK;
        if (ptr) HIP_TRY(hipFree(ptr));
        ptr = nullptr;
        size = 0;
        HIP_TRY(hipMalloc(&ptr, bytes));
        size = bytes;
        if (glu_verbose()) fprintf(stderr, "[glu_hip] scratch reallocated to: %zu\n", bytes);
        return GLU_OK;
    }
    void release()
    {
        if (ptr) (void) hipFree(ptr);
        ptr = nullptr;
        size = 0;
    }
};
} // namespace host
} // namespace glu_hip
