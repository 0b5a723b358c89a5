// scan_stream_bench.hip -- round 4 tuning harness for the chained (single-pass) exclusive scan (not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -o tools/scan_stream_bench tools/scan_stream_bench.hip
//   ./tools/scan_stream_bench [log2n]
// Times, on the same array: the one-workgroup-per-chunk chained kernel (round 2/3), the carry-free kernel of many small
// partitions (the ceiling of an in-place read + write of this shape), and the persistent double-buffered kernel in several
// shapes (16-byte groups per buffer, register budget, workgroups per CU).  Every variant's output is compared with the
// first one's bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "scan_reduce_kernels.hpp"

using namespace glu_hip;

#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                                                       \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

using U = Elem<uint32_t, 1>;

template<int GROUPS, int MINW>
__global__ __launch_bounds__(kChainThreads, MINW) void stream_variant(U* data, uint64_t count, uint32_t chunks, uint32_t total,
                                                                      unsigned long long* chain, uint32_t* ticket, uint32_t epoch)
{
    scan_chained_stream_body<uint32_t, 1, true, GROUPS>(data, count, chunks, total, chain, ticket, epoch);
}

__global__ void fill(uint32_t* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        p[i] = (uint32_t) (i * 2654435761u) >> 20;
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t) 1 << log2n;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *src, *work, *ref;
    unsigned long long* chain;
    uint32_t* ticket;
    CK(hipMalloc(&src, n * 4));
    CK(hipMalloc(&work, n * 4));
    CK(hipMalloc(&ref, n * 4));
    CK(hipMalloc(&chain, (n / 4096 + 16) * 8));
    CK(hipMalloc(&ticket, 256));
    CK(hipMemset(chain, 0, (n / 4096 + 16) * 8));
    hipLaunchKernelGGL(fill, dim3(cus * 8), dim3(256), 0, 0, src, n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t epoch = 0;
    bool have_ref = false;
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0;
        const int reps = 7;
        for (int r = 0; r < reps; r++)
        {
            CK(hipMemcpy(work, src, n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemset(ticket, 0, 16));
            epoch++;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = std::min(best, ms), sum += ms;
        }
        const char* verdict = "";
        if (!have_ref)
        {
            CK(hipMemcpy(ref, work, n * 4, hipMemcpyDeviceToDevice));
            have_ref = true;
            // spot-check the reference itself on the host
            std::vector<uint32_t> a(1 << 20), b(1 << 20);
            CK(hipMemcpy(a.data(), src, a.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
            uint32_t acc = 0;
            bool ok = true;
            for (size_t i = 0; i < a.size() && i < n; i++) ok = ok && b[i] == acc, acc += a[i];
            verdict = ok ? "(host check of the first 2^20 elements: ok)" : "(HOST CHECK FAILED)";
        }
        else if (name[0] != '~')
        {
            // compare with the reference on the device: a tiny reduction on the host of a sampled comparison is enough here
            std::vector<uint32_t> a(n > (1u << 24) ? (1u << 24) : n), b(a.size());
            const size_t off = n - a.size();
            CK(hipMemcpy(a.data(), ref + off, a.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), work + off, b.size() * 4, hipMemcpyDeviceToHost));
            verdict = a == b ? "(same bits as the first variant over the last 2^24 elements)" : "(DIFFERS FROM THE FIRST VARIANT)";
        }
        printf("%-58s best %.4f ms  mean %.4f ms  %.0f GB/s  %s\n", name, best, sum / (reps - 1), n * 8.0 / best / 1e6, verdict);
        fflush(stdout);
    };
    using C8 = ScanCfg<U, kChainGroups, kChainThreads>;
    const uint32_t chunks8 = (uint32_t) ((n + C8::CHUNK - 1) / C8::CHUNK);
    run("one workgroup per chunk (1024 x 8 groups), tickets", [&] {
        hipLaunchKernelGGL((scan_chunks_kernel<uint32_t, 1, true, true>), dim3(chunks8), dim3(kChainThreads), 0, 0, (U*) work, (const U*) nullptr,
                           (uint64_t) n, chunks8, chain, ticket, epoch);
    });
    run("~ no carry: 2^(n-8) partitions of 256 (ceiling, other result)", [&] {
        hipLaunchKernelGGL((scan_small_partitions_kernel<uint32_t, 1, true>), dim3((uint32_t) (n / ScanCfg<U>::CHUNK)), dim3(256), 0, 0, (U*) work,
                           (uint64_t) n, 256u);
    });
#define STREAM(G, MINW, WGS)                                                                                                     \
    {                                                                                                                            \
        using CS = ScanCfg<U, G, kChainThreads>;                                                                                 \
        const uint32_t ch = (uint32_t) ((n + CS::CHUNK - 1) / CS::CHUNK);                                                        \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "stream: %d groups/buffer, min waves/EU %d, %d workgroups/CU", G, MINW, WGS);                 \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL((stream_variant<G, MINW>), dim3(std::min<uint32_t>(ch, cus* WGS)), dim3(kChainThreads), 0, 0, (U*) work, \
                               (uint64_t) n, ch, ch, chain, ticket, epoch);                                                      \
        });                                                                                                                      \
    }
    STREAM(4, 4, 1)
    STREAM(4, 4, 2)
    STREAM(4, 8, 2)
    STREAM(4, 8, 1)
    STREAM(2, 8, 2)
    STREAM(2, 4, 2)
    STREAM(8, 4, 1)
    STREAM(6, 4, 1)
    STREAM(3, 8, 2)
    return 0;
}
