// contiguous_pattern_bench.hip -- round 4: does PHYSICALLY CONTIGUOUS memory (hipExtMallocWithFlags hipDeviceMallocContiguous)
// make the distance between the key scratch and the value scratch a choice instead of a draw?  Earlier rounds cut both arrays
// from one ordinary allocation and found that virtual offsets do not pin physical phases; with a
// physically contiguous block the virtual distance IS the physical one.  The line scatter's access pattern (no sorting work,
// whole 128-byte line stores, 256 regions): caller pair (two plain hipMallocs) -> scratch pair and back, the scratch pair cut
// from ONE block as keys at 0 and values at 1 GiB + delta, for a contiguous block and for an ordinary one.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/contiguous_pattern_bench tools/contiguous_pattern_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void store_nt(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }

__global__ __launch_bounds__(1024) void line_pattern_kernel(const u32x4* __restrict__ ka, const u32x4* __restrict__ va,
                                                            u32x4* __restrict__ kb, u32x4* __restrict__ vb, uint32_t per_wg,
                                                            uint32_t region_len)
{
    constexpr uint32_t RUN = 32, VTILE = 256u * RUN, STEP = 1024u * 4u;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t vtiles = per_wg / VTILE;
    const size_t base = (size_t) b * per_wg;
    for (uint32_t x0 = 0; x0 + 2 * STEP <= vtiles * VTILE; x0 += 2 * STEP)
    {
        u32x4 r[2][2];
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const size_t e = base + x0 + h * STEP + tid * 4u;
            r[h][0] = ka[e / 4];
            r[h][1] = va[e / 4];
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const uint32_t x = x0 + h * STEP + tid * 4u;
            const uint32_t vt = x / VTILE, w = x - vt * VTILE;
            const uint32_t c = w / RUN, o = w - c * RUN;
            const uint32_t j = (c + b * 37u + vt * 11u) & 255u;
            const size_t d = (size_t) j * region_len + ((size_t) b * vtiles + vt) * RUN + o;
            store_nt(&kb[d / 4], r[h][0]);
            store_nt(&vb[d / 4], r[h][1]);
        }
    }
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t) 1 << log2n, bytes = n * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256;
    const uint32_t per_wg = (uint32_t) (n / wgs), region_len = (uint32_t) (n / 256);
    auto time_it = [&](const void* ka, const void* va, void* kb, void* vb) {
        float best = 1e9f;
        for (int r = 0; r < 3; r++)
        {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(line_pattern_kernel, dim3(wgs), dim3(1024), 0, 0, (const u32x4*) ka, (const u32x4*) va, (u32x4*) kb, (u32x4*) vb,
                               per_wg, region_len);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = std::min(best, ms);
        }
        return best;
    };
    void *ck, *cv;
    CK(hipMalloc(&ck, bytes + (1 << 20)));
    CK(hipMalloc(&cv, bytes + (1 << 20)));
    CK(hipMemset(ck, 1, bytes));
    CK(hipMemset(cv, 2, bytes));
    const std::vector<size_t> deltas_mib = {0, 64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024};
    const size_t block = 2 * bytes + ((size_t) 1100 << 20);
    for (int contiguous = 1; contiguous >= 0; contiguous--)
    {
        unsigned char* blk = nullptr;
        hipError_t e = contiguous ? hipExtMallocWithFlags((void**) &blk, block, hipDeviceMallocContiguous) : hipMalloc((void**) &blk, block);
        if (e != hipSuccess)
        {
            printf("%s block of %zu MiB: %s\n", contiguous ? "contiguous" : "ordinary", block >> 20, hipGetErrorString(e));
            (void) hipGetLastError();
            continue;
        }
        printf("%s block of %zu MiB at %p; caller pair at %p / %p\n", contiguous ? "PHYSICALLY CONTIGUOUS" : "ordinary hipMalloc", block >> 20,
               (void*) blk, ck, cv);
        printf("%10s %12s %12s %10s\n", "delta MiB", "caller->S ms", "S->caller ms", "sum x2");
        for (size_t dm : deltas_mib)
        {
            void* sk = blk;
            void* sv = blk + bytes + (dm << 20);
            const float a = time_it(ck, cv, sk, sv), b = time_it(sk, sv, ck, cv);
            printf("%10zu %12.3f %12.3f %10.3f\n", dm, a, b, 2 * (a + b));
        }
        CK(hipFree(blk));
    }
    // the caller pair itself as a block: what a caller who allocates BOTH its arrays in one contiguous block would get
    {
        unsigned char *cblk = nullptr, *sblk = nullptr;
        if (hipExtMallocWithFlags((void**) &cblk, block, hipDeviceMallocContiguous) == hipSuccess &&
            hipExtMallocWithFlags((void**) &sblk, block, hipDeviceMallocContiguous) == hipSuccess)
        {
            CK(hipMemset(cblk, 1, block));
            printf("both pairs from contiguous blocks (caller values at 1 GiB + dc, scratch values at 1 GiB + ds):\n");
            for (size_t dc : {(size_t) 0, (size_t) 256, (size_t) 512, (size_t) 768})
                for (size_t ds : {(size_t) 0, (size_t) 256, (size_t) 512, (size_t) 768})
                {
                    const float a = time_it(cblk, cblk + bytes + (dc << 20), sblk, sblk + bytes + (ds << 20));
                    const float b = time_it(sblk, sblk + bytes + (ds << 20), cblk, cblk + bytes + (dc << 20));
                    printf("  dc %4zu ds %4zu: %.3f + %.3f ms\n", dc, ds, a, b);
                }
        }
        else
            (void) hipGetLastError();
    }
    return 0;
}
