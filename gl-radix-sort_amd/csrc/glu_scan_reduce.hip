// glu_scan_reduce.hip -- BlellochScan and Reduce of libglu_hip.so (the C ABI of include/glu_hip.h; kernels: scan_reduce_kernels.hpp).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "glu_host.hpp"
#include "scan_reduce_kernels.hpp"

using namespace glu_hip;
using namespace glu_hip::host;

// ------------------------------------------------------------------------------------------------------------
// scan / reduce: data-type dispatch
// ------------------------------------------------------------------------------------------------------------
namespace
{
size_t data_type_size(glu_data_type t)
{
    switch (t)
    {
    case GLU_DATA_TYPE_FLOAT: case GLU_DATA_TYPE_INT: case GLU_DATA_TYPE_UINT: return 4;
    case GLU_DATA_TYPE_DOUBLE: case GLU_DATA_TYPE_VEC2: case GLU_DATA_TYPE_UVEC2: case GLU_DATA_TYPE_IVEC2: return 8;
    case GLU_DATA_TYPE_VEC4: case GLU_DATA_TYPE_UVEC4: case GLU_DATA_TYPE_IVEC4: case GLU_DATA_TYPE_DVEC2: return 16;
    case GLU_DATA_TYPE_DVEC4: return 32;
    default: return 0;
    }
}

// calls f.template operator()<S, N>() for the scalar type / component count of `t`
template<typename F>
glu_status dispatch_type(glu_data_type t, F&& f)
{
    switch (t)
    {
    case GLU_DATA_TYPE_FLOAT: return f.template operator()<float, 1>();
    case GLU_DATA_TYPE_DOUBLE: return f.template operator()<double, 1>();
    case GLU_DATA_TYPE_INT: return f.template operator()<int32_t, 1>();
    case GLU_DATA_TYPE_UINT: return f.template operator()<uint32_t, 1>();
    case GLU_DATA_TYPE_VEC2: return f.template operator()<float, 2>();
    case GLU_DATA_TYPE_VEC4: return f.template operator()<float, 4>();
    case GLU_DATA_TYPE_DVEC2: return f.template operator()<double, 2>();
    case GLU_DATA_TYPE_DVEC4: return f.template operator()<double, 4>();
    case GLU_DATA_TYPE_UVEC2: return f.template operator()<uint32_t, 2>();
    case GLU_DATA_TYPE_UVEC4: return f.template operator()<uint32_t, 4>();
    case GLU_DATA_TYPE_IVEC2: return f.template operator()<int32_t, 2>();
    case GLU_DATA_TYPE_IVEC4: return f.template operator()<int32_t, 4>();
    default: return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid data type: %d", (int) t);
    }
}
} // namespace

struct glu_scan_s
{
    glu_data_type type;
    Scratch sums;
    // chained (single-pass) scan state for 4-byte element types: one 64-bit word per chunk + a ticket counter
    Scratch chain;
    Scratch ticket;
    uint32_t epoch = 0;
    bool chained = true; // GLU_HIP_SCAN_CHAINED=0 falls back to reduce-then-scan
    size_t chain_min_chunks = kChainMinChunks; // GLU_HIP_SCAN_CHAINED=2: chained from 2 chunks up (tests)
};

struct glu_reduce_s
{
    glu_data_type type;
    glu_reduce_operator op;
    Scratch partials;
};

namespace
{
constexpr int kReduceMaxBlocks = 8192;     // size of the partials buffer
constexpr int kReduceDefaultBlocks = 512;  // first-stage grid for large inputs: 2 x 256 threads per CU, 64 B in flight
                                           // per lane, measured 6.4 TB/s at 2^28 uint32 (1024-8192 workgroups: 5.3-5.6 TB/s)

// number of chunk-sum elements over all recursion levels
template<typename T>
size_t scan_scratch_elems(size_t count, size_t partitions)
{
    size_t total = 0;
    size_t c = count;
    while (c > (size_t) ScanCfg<T>::CHUNK)
    {
        c = (c + ScanCfg<T>::CHUNK - 1) / ScanCfg<T>::CHUNK;
        total += c * partitions;
    }
    return total;
}

// single-pass chained scan (4-byte element types, more than one chunk per partition)
template<typename S, int N>
glu_status scan_chained(glu_scan_s* scan, Elem<S, N>* data, size_t count, size_t partitions, hipStream_t stream)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T, kChainGroups, kChainThreads>;
    const size_t chunks = (count + C::CHUNK - 1) / C::CHUNK;
    const size_t words = chunks * partitions;
    if (scan->chain.size < words * sizeof(unsigned long long))
    {
        GLU_TRY(scan->chain.reserve(words * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(scan->chain.ptr, 0, scan->chain.size, stream)); // epoch 0 = never ready
        scan->epoch = 0;
    }
    GLU_TRY(scan->ticket.reserve(256));
    if (++scan->epoch >= (1u << 30))
    {
        HIP_TRY(hipMemsetAsync(scan->chain.ptr, 0, scan->chain.size, stream));
        scan->epoch = 1;
    }
    HIP_TRY(hipMemsetAsync(scan->ticket.ptr, 0, 16, stream));
    const bool aligned = ((uintptr_t) data % 16 == 0) && (partitions == 1 || (count * sizeof(T)) % 16 == 0);
    const dim3 grid((uint32_t) words);
    if (aligned)
        hipLaunchKernelGGL((scan_chunks_kernel<S, N, true, true>), grid, dim3(C::THREADS), 0, stream, data, (const T*) nullptr,
                           (uint64_t) count, (uint32_t) chunks, (unsigned long long*) scan->chain.ptr,
                           (uint32_t*) scan->ticket.ptr, scan->epoch);
    else
        hipLaunchKernelGGL((scan_chunks_kernel<S, N, false, true>), grid, dim3(C::THREADS), 0, stream, data, (const T*) nullptr,
                           (uint64_t) count, (uint32_t) chunks, (unsigned long long*) scan->chain.ptr,
                           (uint32_t*) scan->ticket.ptr, scan->epoch);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

template<typename S, int N>
glu_status scan_level(Elem<S, N>* data, size_t count, size_t partitions, Elem<S, N>* scratch, hipStream_t stream)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T>;
    const size_t chunks = (count + C::CHUNK - 1) / C::CHUNK;
    if (chunks * partitions > 0x7FFFFFFFull) return fail(GLU_ERROR_INVALID_ARGUMENT, "scan too large");
    const bool aligned = ((uintptr_t) data % 16 == 0) && (partitions == 1 || (count * sizeof(T)) % 16 == 0);
    const dim3 grid((uint32_t) (chunks * partitions));
    // many small partitions: a workgroup takes CHUNK consecutive elements = several whole partitions (the array is one
    // contiguous run of partitions, so only the array's own alignment matters)
    if (partitions >= 2 && count <= (size_t) C::WAVE_ELEMS && (count & (count - 1)) == 0 && partitions * count > (size_t) C::CHUNK)
    {
        const uint64_t total = (uint64_t) partitions * count;
        const dim3 sgrid((uint32_t) ((total + C::CHUNK - 1) / C::CHUNK));
        if ((uintptr_t) data % 16 == 0)
            hipLaunchKernelGGL((scan_small_partitions_kernel<S, N, true>), sgrid, dim3(C::THREADS), 0, stream, data, total, (uint32_t) count);
        else
            hipLaunchKernelGGL((scan_small_partitions_kernel<S, N, false>), sgrid, dim3(C::THREADS), 0, stream, data, total, (uint32_t) count);
        HIP_TRY(hipGetLastError());
        return GLU_OK;
    }
    if (chunks == 1)
    {
        if (aligned)
            hipLaunchKernelGGL((scan_chunks_kernel<S, N, true>), grid, dim3(C::THREADS), 0, stream, data, (const T*) nullptr, (uint64_t) count, 1u);
        else
            hipLaunchKernelGGL((scan_chunks_kernel<S, N, false>), grid, dim3(C::THREADS), 0, stream, data, (const T*) nullptr, (uint64_t) count, 1u);
        HIP_TRY(hipGetLastError());
        return GLU_OK;
    }
    T* sums = scratch;
    if (aligned)
        hipLaunchKernelGGL((scan_chunk_sums_kernel<S, N, true>), grid, dim3(C::THREADS), 0, stream, (const T*) data, sums, (uint64_t) count, (uint32_t) chunks);
    else
        hipLaunchKernelGGL((scan_chunk_sums_kernel<S, N, false>), grid, dim3(C::THREADS), 0, stream, (const T*) data, sums, (uint64_t) count, (uint32_t) chunks);
    HIP_TRY(hipGetLastError()); // every launch is checked where it happens
    GLU_TRY((scan_level<S, N>(sums, chunks, partitions, scratch + chunks * partitions, stream)));
    if (aligned)
        hipLaunchKernelGGL((scan_chunks_kernel<S, N, true>), grid, dim3(C::THREADS), 0, stream, data, (const T*) sums, (uint64_t) count, (uint32_t) chunks);
    else
        hipLaunchKernelGGL((scan_chunks_kernel<S, N, false>), grid, dim3(C::THREADS), 0, stream, data, (const T*) sums, (uint64_t) count, (uint32_t) chunks);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

struct ScanRunner
{
    glu_scan_s* scan;
    void* data;
    size_t count, partitions;
    hipStream_t stream;
    bool size_only;
    template<typename S, int N>
    glu_status operator()()
    {
        using T = Elem<S, N>;
        if constexpr (sizeof(T) == 4)
        {
            const size_t chunks = (count + ScanCfg<T, kChainGroups, kChainThreads>::CHUNK - 1) / ScanCfg<T, kChainGroups, kChainThreads>::CHUNK;
            // Below about one chunk per CU the ticket chain is latency-bound and the three-launch reduce-then-scan
            // wins (measured crossover between 2^22 and 2^24 elements, tools/scan_probe.py).
            // A captured launch would bake this call's epoch into the graph: every replay would accept the chain words of the
            // replay before as ready.  Under stream capture the scan takes the reduce-then-scan path (capturable: no host state).
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (!size_only) (void) hipStreamIsCapturing(stream, &capturing);
            if (scan->chained && chunks > 1 && chunks * partitions >= scan->chain_min_chunks && chunks * partitions <= 0x7FFFFFFFull &&
                capturing == hipStreamCaptureStatusNone)
            {
                if (size_only)
                {
                    GLU_TRY(scan->ticket.reserve(256));
                    if (scan->chain.size < chunks * partitions * 8)
                    {
                        GLU_TRY(scan->chain.reserve(chunks * partitions * 8));
                        HIP_TRY(hipMemset(scan->chain.ptr, 0, scan->chain.size));
                        scan->epoch = 0;
                    }
                    // no return: a captured run of the same scan takes the other path and needs its scratch too
                }
                else
                    return scan_chained<S, N>(scan, (T*) data, count, partitions, stream);
            }
        }
        size_t need = scan_scratch_elems<T>(count, partitions) * sizeof(T);
        if (need) GLU_TRY(scan->sums.reserve(need));
        if (size_only) return GLU_OK;
        return scan_level<S, N>((T*) data, count, partitions, (T*) scan->sums.ptr, stream);
    }
};

template<int OP, typename S, int N>
glu_status reduce_launch(glu_reduce_s* r, Elem<S, N>* data, size_t count, hipStream_t stream)
{
    using T = Elem<S, N>;
    const bool aligned = ((uintptr_t) data % 16) == 0;
    const size_t vec = (aligned && sizeof(T) < 16) ? 16 / sizeof(T) : 1;
    const size_t packs = count / vec;
    static const size_t max_blocks = [] {
        const char* e = glu_env("GLU_HIP_REDUCE_BLOCKS"); // tuning override
        const long v = e ? atol(e) : 0;
        return (size_t) (v > 0 && v <= kReduceMaxBlocks ? v : kReduceDefaultBlocks);
    }();
    size_t blocks = std::max<size_t>(1, std::min<size_t>(packs / (256 * 4), max_blocks));
    T* partials = (T*) r->partials.ptr;
    if (blocks == 1)
    {
        if (aligned) hipLaunchKernelGGL((reduce_kernel<OP, S, N, true>), dim3(1), dim3(256), 0, stream, (const T*) data, data, (uint64_t) count);
        else hipLaunchKernelGGL((reduce_kernel<OP, S, N, false>), dim3(1), dim3(256), 0, stream, (const T*) data, data, (uint64_t) count);
    }
    else
    {
        if (aligned) hipLaunchKernelGGL((reduce_kernel<OP, S, N, true>), dim3((uint32_t) blocks), dim3(256), 0, stream, (const T*) data, partials, (uint64_t) count);
        else hipLaunchKernelGGL((reduce_kernel<OP, S, N, false>), dim3((uint32_t) blocks), dim3(256), 0, stream, (const T*) data, partials, (uint64_t) count);
        hipLaunchKernelGGL((reduce_kernel<OP, S, N, true>), dim3(1), dim3(256), 0, stream, (const T*) partials, data, (uint64_t) blocks);
    }
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

struct ReduceRunner
{
    glu_reduce_s* red;
    void* data;
    size_t count;
    hipStream_t stream;
    template<typename S, int N>
    glu_status operator()()
    {
        using T = Elem<S, N>;
        switch (red->op)
        {
        case GLU_REDUCE_SUM: return reduce_launch<OP_SUM, S, N>(red, (T*) data, count, stream);
        case GLU_REDUCE_MUL: return reduce_launch<OP_MUL, S, N>(red, (T*) data, count, stream);
        case GLU_REDUCE_MIN: return reduce_launch<OP_MIN, S, N>(red, (T*) data, count, stream);
        case GLU_REDUCE_MAX: return reduce_launch<OP_MAX, S, N>(red, (T*) data, count, stream);
        default: return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid reduction operator: %d", (int) red->op);
        }
    }
};
} // namespace

extern "C" {

glu_status glu_scan_create(glu_data_type data_type, glu_scan* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    if ((int) data_type < 0 || data_type >= GLU_DATA_TYPE_COUNT_)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid data type: %d", (int) data_type);
    glu_scan_s* s = new glu_scan_s();
    s->type = data_type;
    if (const char* e = glu_env("GLU_HIP_SCAN_CHAINED"))
    {
        s->chained = atoi(e) != 0;
        if (atoi(e) == 2) s->chain_min_chunks = 2;
    }
    *out = s;
    return GLU_OK;
}

glu_status glu_scan_destroy(glu_scan scan)
{
    GLU_TRY(enter());
    if (!scan) return GLU_OK;
    (void) hipDeviceSynchronize(); // (a caller stream may still run its kernels)
    scan->sums.release();
    scan->chain.release();
    scan->ticket.release();
    delete scan;
    return GLU_OK;
}

glu_status glu_scan_prepare(glu_scan scan, size_t count, size_t num_partitions)
{
    GLU_TRY(enter());
    if (!scan) return fail(GLU_ERROR_INVALID_ARGUMENT, "scan is NULL");
    if (count == 0 || num_partitions == 0) return GLU_OK;
    ScanRunner r{scan, nullptr, count, num_partitions, nullptr, true};
    return dispatch_type(scan->type, r);
}

glu_status glu_scan_run_ptr(glu_scan scan, void* data, size_t count, size_t num_partitions, void* stream)
{
    GLU_TRY(enter());
    if (!scan) return fail(GLU_ERROR_INVALID_ARGUMENT, "scan is NULL");
    if (!data) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid buffer");
    if (count == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Count must be greater than zero");
    if (num_partitions < 1) return fail(GLU_ERROR_INVALID_ARGUMENT, "Num of partitions must be >= 1");
    if (((uintptr_t) data % data_type_size(scan->type)) != 0)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "data is not aligned to its element size");
    ScanRunner r{scan, data, count, num_partitions, pick_stream(stream), false};
    return dispatch_type(scan->type, r);
}

glu_status glu_scan_run(glu_scan scan, glu_buffer buffer, size_t count, size_t num_partitions)
{
    GLU_TRY(enter());
    if (!scan) return fail(GLU_ERROR_INVALID_ARGUMENT, "scan is NULL");
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));                                                         // BlellochScan.hpp:132
    if (count == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Count must be greater than zero");    // :133
    if ((count & (count - 1)) != 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Count must be a power of 2"); // :134
    if (num_partitions < 1) return fail(GLU_ERROR_INVALID_ARGUMENT, "Num of partitions must be >= 1");      // :135
    const size_t es = data_type_size(scan->type);
    if (count > b.size / es / num_partitions)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "count * num_partitions exceeds the buffer size");
    return glu_scan_run_ptr(scan, b.ptr, count, num_partitions, nullptr);
}

glu_status glu_reduce_create(glu_data_type data_type, glu_reduce_operator op, glu_reduce* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    if ((int) data_type < 0 || data_type >= GLU_DATA_TYPE_COUNT_)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid data type: %d", (int) data_type);
    if ((int) op < 0 || op >= GLU_REDUCE_COUNT_)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid reduction operator: %d", (int) op); // Reduce.hpp:94-97
    glu_reduce_s* r = new glu_reduce_s();
    r->type = data_type;
    r->op = op;
    glu_status st = r->partials.reserve((size_t) kReduceMaxBlocks * 32);
    if (st != GLU_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return GLU_OK;
}

glu_status glu_reduce_destroy(glu_reduce reduce)
{
    GLU_TRY(enter());
    if (!reduce) return GLU_OK;
    (void) hipDeviceSynchronize();
    reduce->partials.release();
    delete reduce;
    return GLU_OK;
}

glu_status glu_reduce_run_ptr(glu_reduce reduce, void* data, size_t count, void* stream)
{
    GLU_TRY(enter());
    if (!reduce) return fail(GLU_ERROR_INVALID_ARGUMENT, "reduce is NULL");
    if (!data) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid buffer");
    if (count == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Count must be greater than zero");
    if (((uintptr_t) data % std::min<size_t>(16, data_type_size(reduce->type))) != 0)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "data is not aligned to its element size");
    ReduceRunner r{reduce, data, count, pick_stream(stream)};
    return dispatch_type(reduce->type, r);
}

glu_status glu_reduce_run(glu_reduce reduce, glu_buffer buffer, size_t count)
{
    GLU_TRY(enter());
    if (!reduce) return fail(GLU_ERROR_INVALID_ARGUMENT, "reduce is NULL");
    Buffer b;
    GLU_TRY(lookup(buffer, b, "buffer"));                                                      // Reduce.hpp:113
    if (count == 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "Count must be greater than zero"); // Reduce.hpp:114
    if (count > b.size / data_type_size(reduce->type))
        return fail(GLU_ERROR_INVALID_ARGUMENT, "count exceeds the buffer size");
    return glu_reduce_run_ptr(reduce, b.ptr, count, nullptr);
}

} // extern "C"
