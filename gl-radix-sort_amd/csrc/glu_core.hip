// glu_core.hip -- libglu_hip.so's errors, device, buffers and timer (the C ABI of include/glu_hip.h; gfx950 only, no CPU fallback),
// and the definitions behind glu_host.hpp.
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "glu_host.hpp"

using namespace glu_hip::host;

namespace glu_hip
{
namespace host
{
thread_local std::string g_last_error;

bool glu_verbose()
{
    static const bool on = getenv("GLU_VERBOSE") != nullptr;
    return on;
}

const char* glu_env(const char* name) { return getenv(name); }

glu_status fail(glu_status code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}


Device g_dev;

glu_status ensure_device()
{
    std::lock_guard<std::mutex> lock(g_dev.mutex);
    if (g_dev.ready) return GLU_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(GLU_ERROR_NO_DEVICE,
                    "no HIP device visible (%s): libglu_hip has no CPU fallback, an MI355X (gfx950) is required",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    int id = 0;
    if (g_dev.requested >= 0)
        id = g_dev.requested;
    else
        HIP_TRY(hipGetDevice(&id));
    if (id >= count) return fail(GLU_ERROR_INVALID_ARGUMENT, "device %d does not exist (%d visible)", id, count);
    HIP_TRY(hipSetDevice(id));
    HIP_TRY(hipGetDeviceProperties(&g_dev.props, id));
    if (strncmp(g_dev.props.gcnArchName, "gfx950", 6) != 0)
        return fail(GLU_ERROR_NO_DEVICE, "device %d is %s; libglu_hip is built for gfx950 only", id,
                    g_dev.props.gcnArchName);
    g_dev.id = id;
    g_dev.num_cus = g_dev.props.multiProcessorCount > 0 ? g_dev.props.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&g_dev.queue, hipStreamNonBlocking));
    g_dev.ready = true;
    return GLU_OK;
}

glu_status enter()
{
    GLU_TRY(ensure_device());
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != g_dev.id) HIP_TRY(hipSetDevice(g_dev.id));
    return GLU_OK;
}

namespace
{
std::mutex g_buf_mutex;
std::unordered_map<glu_buffer, Buffer> g_buffers;
glu_buffer g_next_buffer = 1;
} // namespace

glu_status lookup(glu_buffer h, Buffer& out, const char* what)
{
    if (h == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid %s", what);
    std::lock_guard<std::mutex> lock(g_buf_mutex);
    auto it = g_buffers.find(h);
    if (it == g_buffers.end()) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid %s (unknown handle %u)", what, h);
    out = it->second;
    return GLU_OK;
}

namespace
{
glu_buffer register_buffer(const Buffer& b)
{
    std::lock_guard<std::mutex> lock(g_buf_mutex);
    glu_buffer h = g_next_buffer++;
    if (g_next_buffer == 0) g_next_buffer = 1;
    g_buffers[h] = b;
    return h;
}
} // namespace
} // namespace host
} // namespace glu_hip

// ------------------------------------------------------------------------------------------------------------
// library / device
// ------------------------------------------------------------------------------------------------------------
extern "C" {

const char* glu_last_error(void) { return g_last_error.c_str(); }
const char* glu_version(void) { return "glu_hip 0.6.0 gfx950"; }

glu_status glu_device_count(int* count)
{
    if (!count) return fail(GLU_ERROR_INVALID_ARGUMENT, "count is NULL");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return GLU_OK;
}

glu_status glu_set_device(int device)
{
    {
        std::lock_guard<std::mutex> lock(g_dev.mutex);
        if (g_dev.ready)
        {
            if (device == g_dev.id) return GLU_OK;
            return fail(GLU_ERROR_INVALID_STATE, "device already initialised to %d (one device per process)", g_dev.id);
        }
        g_dev.requested = device;
    }
    return ensure_device();
}

glu_status glu_device_info(char* out, size_t out_size)
{
    GLU_TRY(enter());
    if (!out || out_size == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    snprintf(out, out_size, "%s (%s), %d CUs, %.1f GiB, device %d", g_dev.props.name, g_dev.props.gcnArchName,
             g_dev.num_cus, (double) g_dev.props.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), g_dev.id);
    return GLU_OK;
}

glu_status glu_device_synchronize(void)
{
    GLU_TRY(enter());
    HIP_TRY(hipStreamSynchronize(g_dev.queue));
    return GLU_OK;
}

glu_status glu_queue(void** stream)
{
    GLU_TRY(enter());
    if (!stream) return fail(GLU_ERROR_INVALID_ARGUMENT, "stream is NULL");
    *stream = (void*) g_dev.queue;
    return GLU_OK;
}

// ------------------------------------------------------------------------------------------------------------
// buffers
// ------------------------------------------------------------------------------------------------------------
glu_status glu_buffer_create(size_t size, glu_buffer* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    Buffer b;
    b.size = size;
    if (size > 0) HIP_TRY(hipMalloc(&b.ptr, size));
    *out = register_buffer(b);
    return GLU_OK;
}

glu_status glu_buffer_create_with_data(const void* data, size_t size, glu_buffer* out)
{
    GLU_TRY(enter());
    if (!data) return fail(GLU_ERROR_INVALID_ARGUMENT, "data is NULL");
    if (size == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "size is 0");
    GLU_TRY(glu_buffer_create(size, out));
    const glu_status st = glu_buffer_write(*out, data, size, 0);
    if (st != GLU_OK)
    {
        const std::string message = g_last_error; // (the destroy below must not replace the reason)
        (void) glu_buffer_destroy(*out);
        *out = 0;
        g_last_error = message;
    }
    return st;
}

glu_status glu_buffer_wrap(void* device_ptr, size_t size, glu_buffer* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    if (!device_ptr && size > 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "device_ptr is NULL");
    Buffer b;
    b.ptr = device_ptr;
    b.size = size;
    b.owned = false;
    *out = register_buffer(b);
    return GLU_OK;
}

glu_status glu_buffer_destroy(glu_buffer buffer)
{
    GLU_TRY(enter());
    if (buffer == 0) return GLU_OK;
    Buffer b;
    {
        std::lock_guard<std::mutex> lock(g_buf_mutex);
        auto it = g_buffers.find(buffer);
        if (it == g_buffers.end()) return fail(GLU_ERROR_INVALID_ARGUMENT, "unknown buffer handle %u", buffer);
        b = it->second;
        g_buffers.erase(it);
    }
    if (b.owned && b.ptr)
    {
        // queued work may still use it: free after the queue drains (hipFree synchronises the device anyway)
        HIP_TRY(hipStreamSynchronize(g_dev.queue));
        HIP_TRY(hipFree(b.ptr));
    }
    return GLU_OK;
}

glu_status glu_buffer_size(glu_buffer buffer, size_t* size)
{
    GLU_TRY(enter());
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));
    if (!size) return fail(GLU_ERROR_INVALID_ARGUMENT, "size is NULL");
    *size = b.size;
    return GLU_OK;
}

glu_status glu_buffer_device_ptr(glu_buffer buffer, void** device_ptr)
{
    GLU_TRY(enter());
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));
    if (!device_ptr) return fail(GLU_ERROR_INVALID_ARGUMENT, "device_ptr is NULL");
    *device_ptr = b.ptr;
    return GLU_OK;
}

glu_status glu_buffer_write(glu_buffer buffer, const void* data, size_t size, size_t offset)
{
    GLU_TRY(enter());
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));
    if (size == 0) return GLU_OK;
    if (!data) return fail(GLU_ERROR_INVALID_ARGUMENT, "data is NULL");
    if (offset > b.size || size > b.size - offset)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "write of %zu bytes at %zu exceeds buffer size %zu", size, offset, b.size);
    // pageable host memory: the copy is staged before the call returns, ordered on the queue
    HIP_TRY(hipMemcpyAsync((char*) b.ptr + offset, data, size, hipMemcpyHostToDevice, g_dev.queue));
    HIP_TRY(hipStreamSynchronize(g_dev.queue));
    return GLU_OK;
}

glu_status glu_buffer_read(glu_buffer buffer, void* data, size_t size, size_t offset)
{
    GLU_TRY(enter());
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));
    if (size == 0) return GLU_OK;
    if (!data) return fail(GLU_ERROR_INVALID_ARGUMENT, "data is NULL");
    if (offset > b.size || size > b.size - offset)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "read of %zu bytes at %zu exceeds buffer size %zu", size, offset, b.size);
    HIP_TRY(hipMemcpyAsync(data, (const char*) b.ptr + offset, size, hipMemcpyDeviceToHost, g_dev.queue));
    HIP_TRY(hipStreamSynchronize(g_dev.queue));
    return GLU_OK;
}

glu_status glu_buffer_fill_u32(glu_buffer buffer, uint32_t value)
{
    GLU_TRY(enter());
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));
    if (b.size / 4 > 0) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) b.ptr, (int) value, b.size / 4, g_dev.queue));
    return GLU_OK;
}

glu_status glu_buffer_copy(glu_buffer src, glu_buffer dst, size_t size, size_t src_offset, size_t dst_offset)
{
    GLU_TRY(enter());
    Buffer s, d;
    GLU_TRY(lookup(src, s, "source buffer"));
    GLU_TRY(lookup(dst, d, "destination buffer"));
    if (src_offset > s.size || size > s.size - src_offset || dst_offset > d.size || size > d.size - dst_offset)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "copy of %zu bytes out of range", size);
    if (size > 0)
        HIP_TRY(hipMemcpyAsync((char*) d.ptr + dst_offset, (const char*) s.ptr + src_offset, size,
                               hipMemcpyDeviceToDevice, g_dev.queue));
    return GLU_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// timer
// ------------------------------------------------------------------------------------------------------------
struct glu_timer_s
{
    hipEvent_t start, stop;
};

extern "C" {

glu_status glu_timer_begin(glu_timer* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    glu_timer_s* t = new glu_timer_s();
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess)
    {
        delete t;
        return fail(GLU_ERROR_DEVICE, "hipEventCreate failed");
    }
    HIP_TRY(hipEventRecord(t->start, g_dev.queue));
    *out = t;
    return GLU_OK;
}

glu_status glu_timer_end(glu_timer timer, uint64_t* elapsed_ns)
{
    GLU_TRY(enter());
    if (!timer) return fail(GLU_ERROR_INVALID_ARGUMENT, "timer is NULL");
    HIP_TRY(hipEventRecord(timer->stop, g_dev.queue));
    HIP_TRY(hipEventSynchronize(timer->stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, timer->start, timer->stop));
    if (elapsed_ns) *elapsed_ns = (uint64_t) ((double) ms * 1.0e6);
    (void) hipEventDestroy(timer->start);
    (void) hipEventDestroy(timer->stop);
    delete timer;
    return GLU_OK;
}

} // extern "C"

