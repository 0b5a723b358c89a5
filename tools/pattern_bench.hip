// pattern_bench.hip -- what does the memory system deliver for the scatter's ACCESS PATTERN, with no sorting work at all?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/pattern_bench tools/pattern_bench.hip
// A "chunk copy": workgroup b streams its contiguous range of two input arrays (keys, vals) and writes chunk j of every
// tile (CHUNK consecutive elements) to region j % REGIONS of the two output arrays -- the write pattern of a counting
// pass on uniform keys with REGIONS digit values and CHUNK = TILE / REGIONS elements per (tile, digit) run, 64-byte
// aligned like the carry makes them.  Nothing is ranked or staged, every load and store is independent, so the time is
// the memory system's ceiling for the pattern.  Not part of the product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

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

// Workgroup b owns elements [b * per_wg, (b + 1) * per_wg).  Its range is cut into virtual tiles of REGIONS * CHUNK
// elements; chunk j of virtual tile t goes to out[j * region_len + (b * vtiles_per_wg + t) * CHUNK ...].
template<int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void chunk_copy_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ va,
                                                             uint32_t* __restrict__ kb, uint32_t* __restrict__ vb,
                                                             uint32_t per_wg, uint32_t chunk_shift, uint32_t region_shift,
                                                             uint32_t region_len, int rotate)
{
    constexpr int STEP = THREADS * KPT;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t chunk_mask = (1u << chunk_shift) - 1, regions = 1u << region_shift;
    const uint32_t vt_shift = chunk_shift + region_shift, vtiles_per_wg = per_wg >> vt_shift;
    for (uint32_t x0 = 0; x0 < per_wg; x0 += STEP)
    {
        const size_t src = (size_t) b * per_wg + x0;
        uint32_t k[KPT], v[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++) k[i] = ka[src + i * THREADS + tid];
#pragma unroll
        for (int i = 0; i < KPT; i++) v[i] = va[src + i * THREADS + tid];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t x = x0 + i * THREADS + tid;
            const uint32_t vt = x >> vt_shift, e = x & ((1u << vt_shift) - 1);
            uint32_t j = e >> chunk_shift;
            if (rotate) j = (j + b * 37u + vt * 11u) & (regions - 1); // de-correlate which region the workgroups hit at one time
            const size_t dst = (size_t) j * region_len + ((size_t) b * vtiles_per_wg + vt) * (chunk_mask + 1) + (e & chunk_mask);
            kb[dst] = k[i];
            vb[dst] = v[i];
        }
    }
}

// Same pattern for any chunk length (not only powers of two) and with the whole destination shifted by `shift_elems`
// elements: chunks of 48 elements at 64-byte alignment are what the production scatter (12288-pair tiles, 256 digits,
// 64-byte carry) writes; a 128-byte chunk shifted by 64 bytes straddles two 128-byte lines.
template<int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void chunk_copy_any_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ va,
                                                                 uint32_t* __restrict__ kb, uint32_t* __restrict__ vb,
                                                                 uint32_t per_wg, uint32_t chunk, uint32_t regions,
                                                                 uint32_t region_len, uint32_t shift_elems)
{
    constexpr int STEP = THREADS * KPT;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t vtile = chunk * regions, vtiles_per_wg = per_wg / vtile;
    for (uint32_t x0 = 0; x0 + STEP <= vtiles_per_wg * vtile; x0 += STEP)
    {
        const size_t src = (size_t) b * per_wg + x0;
        uint32_t k[KPT], v[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++) k[i] = ka[src + i * THREADS + tid];
#pragma unroll
        for (int i = 0; i < KPT; i++) v[i] = va[src + i * THREADS + tid];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t x = x0 + i * THREADS + tid;
            const uint32_t vt = x / vtile, e = x - vt * vtile;
            uint32_t j = e / chunk;
            const uint32_t r = e - j * chunk;
            j = (j + b * 37u + vt * 11u) % regions;
            const size_t dst = (size_t) j * region_len + ((size_t) b * vtiles_per_wg + vt) * chunk + r + shift_elems;
            kb[dst] = k[i];
            vb[dst] = v[i];
        }
    }
}

// Tile-ordered ("onesweep") write pattern: tile T of the whole input puts its chunk of region j at j * region_len +
// T * chunk, so a chunk that is not a multiple of 32 elements shares its first and last 128-byte line with the tiles
// T - 1 and T + 1, which another workgroup writes at about the same time.  order 0: tile k * wgs + b (neighbours on
// different XCDs), order 1: groups of 32 consecutive tiles on the 32 CUs of one XCD (workgroup b runs on XCD b % 8), so
// that the two halves of a shared line meet in one L2.
template<int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void tile_order_copy_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ va,
                                                                  uint32_t* __restrict__ kb, uint32_t* __restrict__ vb,
                                                                  uint32_t tiles_per_wg, uint32_t chunk, uint32_t regions,
                                                                  uint32_t region_len, int order)
{
    constexpr int STEP = THREADS * KPT;
    const uint32_t b = blockIdx.x, tid = threadIdx.x, wgs = gridDim.x;
    const uint32_t vtile = chunk * regions; // <= STEP
    for (uint32_t k = 0; k < tiles_per_wg; k++)
    {
        const uint32_t T = order == 0 ? k * wgs + b : (k * 8 + (b & 7)) * (wgs / 8) + (b >> 3);
        const size_t src = (size_t) T * vtile;
        uint32_t kk[KPT], vv[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t e = i * THREADS + tid;
            kk[i] = e < vtile ? ka[src + e] : 0u;
            vv[i] = e < vtile ? va[src + e] : 0u;
        }
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t e = i * THREADS + tid;
            if (e < vtile)
            {
                const uint32_t j = e / chunk, r = e - j * chunk;
                const size_t dst = (size_t) j * region_len + (size_t) T * chunk + r;
                kb[dst] = kk[i];
                vb[dst] = vv[i];
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t) 1 << log2n;
    uint32_t *ka, *va, *kb, *vb;
    CK(hipMalloc(&ka, n * 4));
    CK(hipMalloc(&va, n * 4));
    CK(hipMalloc(&kb, n * 4 + (1 << 20)));
    CK(hipMalloc(&vb, n * 4 + (1 << 20)));
    CK(hipMemset(ka, 1, n * 4));
    CK(hipMemset(va, 2, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int threads, int kpt, int wgs, int chunk_shift, int region_shift, int rotate) {
        const uint32_t per_wg = (uint32_t) (n / wgs);
        const uint32_t region_len = (uint32_t) (n >> region_shift);
        float best = 1e9f;
        for (int r = 0; r < 5; r++)
        {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, ka, va, kb, vb, per_wg, (uint32_t) chunk_shift, (uint32_t) region_shift,
                               region_len, rotate);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("threads %4d kpt %2d wgs %5d | %4d regions x chunks of %5d elems (%6d B) rotate %d: %.3f ms  %.0f GB/s\n", threads, kpt,
               wgs, 1 << region_shift, 1 << chunk_shift, 4 << chunk_shift, rotate, best, n * 16.0 / best / 1e6);
        fflush(stdout);
    };
    for (int rot = 0; rot < 2; rot++)
        for (int cs : {4, 5, 6, 7, 8, 10})
            run(chunk_copy_kernel<1024, 16>, 1024, 16, 256, cs, 8, rot);
    for (int cs : {6, 8, 10}) run(chunk_copy_kernel<1024, 16>, 1024, 16, 256, cs, 4, 1); // 16 regions (4-bit digits)
    // more, smaller workgroups in flight: does concurrency move the ceiling?
    for (int cs : {4, 5, 6, 8})
    {
        run(chunk_copy_kernel<512, 16>, 512, 16, 512, cs, 8, 1);
        run(chunk_copy_kernel<256, 16>, 256, 16, 1024, cs, 8, 1);
        run(chunk_copy_kernel<256, 8>, 256, 8, 2048, cs, 8, 1);
    }
    auto run_any = [&](int chunk, int regions, int shift_elems) {
        const int wgs = 256;
        const uint32_t per_wg = (uint32_t) (n / wgs);
        const uint32_t region_len = (uint32_t) (n / regions) & ~31u;
        float best = 1e9f;
        for (int r = 0; r < 5; r++)
        {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((chunk_copy_any_kernel<1024, 16>), dim3(wgs), dim3(1024), 0, 0, ka, va, kb, vb, per_wg, (uint32_t) chunk,
                               (uint32_t) regions, region_len, (uint32_t) shift_elems);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const uint32_t vtile = chunk * regions;
        const double moved = (double) (per_wg / vtile) * vtile / (1024 * 16) * (1024 * 16) * wgs * 16.0;
        printf("any: %4d regions x chunks of %5d elems (%6d B), destination shifted by %3d B: %.3f ms  %.0f GB/s\n", regions, chunk,
               chunk * 4, shift_elems * 4, best, moved / best / 1e6);
        fflush(stdout);
    };
    for (int ch : {16, 32, 48, 64, 80, 96, 128}) run_any(ch, 256, 0);
    auto run_order = [&](int chunk, int order) {
        const int wgs = 256, regions = 256;
        const uint32_t vtile = chunk * regions;
        const uint32_t tiles_per_wg = (uint32_t) (n / vtile / wgs);
        const uint32_t region_len = ((uint32_t) (n / regions) + 31u) & ~31u;
        float best = 1e9f;
        for (int r = 0; r < 5; r++)
        {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((tile_order_copy_kernel<1024, 12>), dim3(wgs), dim3(1024), 0, 0, ka, va, kb, vb, tiles_per_wg, (uint32_t) chunk,
                               (uint32_t) regions, region_len, order);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const double moved = (double) tiles_per_wg * wgs * vtile * 16.0;
        printf("tile order %s: 256 regions x chunks of %3d elems (%4d B): %.3f ms  %.0f GB/s\n",
               order ? "XCD groups of 32" : "round robin     ", chunk, chunk * 4, best, moved / best / 1e6);
        fflush(stdout);
    };
    for (int ch : {32, 36, 40, 44, 48})
        for (int order : {0, 1}) run_order(ch, order);
    return 0;
}
