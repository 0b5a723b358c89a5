// scatter_bench.hip -- tuning harness for the radix-sort kernels (not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -o tools/scatter_bench tools/scatter_bench.hip
//   ./tools/scatter_bench [log2n]
// Times one counting pass (count + row scan + scatter) for several geometries, next to a plain copy kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "radix_sort_kernels.hpp"
#include "radix_scatter_lines.hpp"

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

__global__ void fill_kernel(uint32_t* keys, uint32_t* vals, size_t n, int zero)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t stride = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride)
    {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
        keys[i] = zero ? 0u : (uint32_t) x;
        vals[i] = (uint32_t) i;
    }
}

__global__ void copy_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c,
                            uint4* __restrict__ d, size_t nvec)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t stride = (size_t) gridDim.x * blockDim.x;
    for (; i < nvec; i += stride)
    {
        c[i] = a[i];
        d[i] = b[i];
    }
}

// copy with selectable access widths (bytes per lane) to see what narrow accesses cost
template<int LW, int SW>
__global__ __launch_bounds__(1024) void copy_width_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                          uint32_t* __restrict__ c, uint32_t* __restrict__ d, size_t n)
{
    // each thread moves 16 consecutive dwords of a and of b per iteration
    const size_t chunk = (size_t) blockDim.x * 16;
    for (size_t base = (size_t) blockIdx.x * chunk; base < n; base += (size_t) gridDim.x * chunk)
    {
        uint32_t ra[16], rb[16];
        if (LW == 16)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                uint4 x = *reinterpret_cast<const uint4*>(a + base + (j * blockDim.x + threadIdx.x) * 4);
                uint4 y = *reinterpret_cast<const uint4*>(b + base + (j * blockDim.x + threadIdx.x) * 4);
                ra[4 * j] = x.x; ra[4 * j + 1] = x.y; ra[4 * j + 2] = x.z; ra[4 * j + 3] = x.w;
                rb[4 * j] = y.x; rb[4 * j + 1] = y.y; rb[4 * j + 2] = y.z; rb[4 * j + 3] = y.w;
            }
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                ra[j] = a[base + j * blockDim.x + threadIdx.x];
                rb[j] = b[base + j * blockDim.x + threadIdx.x];
            }
        }
        if (SW == 16)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                *reinterpret_cast<uint4*>(c + base + (j * blockDim.x + threadIdx.x) * 4) = make_uint4(ra[4 * j], ra[4 * j + 1], ra[4 * j + 2], ra[4 * j + 3]);
                *reinterpret_cast<uint4*>(d + base + (j * blockDim.x + threadIdx.x) * 4) = make_uint4(rb[4 * j], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]);
            }
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
                c[base + j * blockDim.x + threadIdx.x] = ra[j];
                d[base + j * blockDim.x + threadIdx.x] = rb[j];
            }
        }
    }
}

// 4-stream copy in which a workgroup never touches a[x] and b[x] (or c[x] and d[x]) close together in time: per chunk it
// first moves a -> c, then b -> d
template<int CH>
__global__ __launch_bounds__(1024) void copy_split_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                          uint32_t* __restrict__ c, uint32_t* __restrict__ d, size_t n)
{
    for (size_t base = (size_t) blockIdx.x * CH; base < n; base += (size_t) gridDim.x * CH)
    {
        for (int i = threadIdx.x; i < CH; i += 16 * 1024)
        {
            uint32_t r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = (i + j * 1024 < CH) ? a[base + i + j * 1024] : 0;
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (i + j * 1024 < CH) c[base + i + j * 1024] = r[j];
        }
        for (int i = threadIdx.x; i < CH; i += 16 * 1024)
        {
            uint32_t r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = (i + j * 1024 < CH) ? b[base + i + j * 1024] : 0;
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (i + j * 1024 < CH) d[base + i + j * 1024] = r[j];
        }
    }
}

__global__ void check_kernel(const uint32_t* keys, const uint32_t* vals, const uint32_t* src_keys, size_t n,
                             uint32_t shift, uint32_t mask, unsigned long long* bad)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t stride = (size_t) gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (; i + 1 < n; i += stride)
    {
        uint32_t d0 = (keys[i] >> shift) & mask, d1 = (keys[i + 1] >> shift) & mask;
        if (d0 > d1) b++;
        if (d0 == d1 && vals[i] > vals[i + 1]) b++; // stable: vals = iota on input
        if (src_keys[vals[i]] != keys[i]) b++;
    }
    if (b) atomicAdd(bad, b);
}

struct Ctx
{
    uint32_t *keys, *vals, *keys2, *vals2, *table;
    unsigned long long* bad;
    size_t n;
    int cus;
    hipEvent_t ev[4];
};

template<typename F>
float time_min(Ctx& c, int reps, F&& f)
{
    float best = 1e30f;
    for (int r = 0; r < reps; r++)
    {
        CK(hipEventRecord(c.ev[0]));
        f();
        CK(hipEventRecord(c.ev[1]));
        CK(hipEventSynchronize(c.ev[1]));
        float ms;
        CK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        best = std::min(best, ms);
    }
    return best;
}

template<int BITS, int THREADS, int KPT, bool CARRY = true, int ABLATE = 0, int ROUNDS = 1, bool PREFETCH = false, bool DMA = false, bool VALS = true>
void run_variant(Ctx& c, int blocks_per_cu, uint32_t shift, uint32_t mask_override = 0)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int TILE = THREADS * KPT;
    using Smem = ScatterSmem<uint32_t, BITS, THREADS, KPT, CARRY, ROUNDS, VALS>;
    const uint32_t tiles = (uint32_t) ((c.n + TILE - 1) / TILE);
    uint32_t nb = std::min<uint32_t>(tiles, (uint32_t) (c.cus * blocks_per_cu));
    if (getenv("SB_NB")) nb = std::min<uint32_t>(tiles, (uint32_t) atoi(getenv("SB_NB"))); // fewer workgroups than CUs: per-CU vs chip limits
    uint32_t* totals = c.table + (size_t) RADIX * nb;
    const uint32_t mask = mask_override ? mask_override : RADIX - 1;
    if (mask_override) printf("mask %u: ", mask);
    auto scatter = radix_scatter_kernel<uint32_t, BITS, THREADS, KPT, CARRY, ABLATE, false, 1, ROUNDS, PREFETCH, DMA, false, VALS>;
    auto scatter_st = radix_scatter_kernel<uint32_t, BITS, THREADS, KPT, CARRY, ABLATE, true, 1, ROUNDS, PREFETCH, DMA, false, VALS>;
    CK(hipFuncSetAttribute((const void*) scatter_st, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));

    float t_count = time_min(c, 5, [&] {
        hipLaunchKernelGGL((radix_count_kernel<uint32_t, BITS, THREADS, TILE>), dim3(nb), dim3(THREADS), 0, 0, c.keys, c.table,
                           (uint32_t) c.n, shift, mask, tiles, 0u);
    });
    float t_scan = time_min(c, 1, [&] {
        hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, 0, c.table, totals, nb);
    });
    float t_scatter = time_min(c, 5, [&] {
        hipLaunchKernelGGL(scatter, dim3(nb), dim3(THREADS), sizeof(Smem), 0, c.keys, VALS ? c.vals : nullptr, c.keys2,
                           VALS ? c.vals2 : nullptr, c.table, totals, (uint32_t) c.n, shift, mask, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
    });
    CK(hipGetLastError());
    if (getenv("SB_TRACE"))
    { // time series of one variant inside one process: clock-state drift shows up here, buffer placement cannot
        printf("    trace:");
        const int groups = atoi(getenv("SB_TRACE")) > 1 ? atoi(getenv("SB_TRACE")) : 60;
        for (int r = 0; r < groups; r++)
        {
            float t = time_min(c, groups > 60 ? 50 : 1, [&] { // long traces: best of 50 launches per point
                hipLaunchKernelGGL(scatter, dim3(nb), dim3(THREADS), sizeof(Smem), 0, c.keys, c.vals, c.keys2, c.vals2, c.table,
                                   totals, (uint32_t) c.n, shift, mask, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
            });
            printf(" %.3f", t);
        }
        printf("\n");
    }
    unsigned long long* st;
    CK(hipMalloc(&st, 64));
    CK(hipMemset(st, 0, 64));
    hipLaunchKernelGGL(scatter_st, dim3(nb), dim3(THREADS), sizeof(Smem), 0, c.keys, VALS ? c.vals : nullptr, c.keys2,
                       VALS ? c.vals2 : nullptr, c.table, totals, (uint32_t) c.n, shift, mask, tiles, st, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
    unsigned long long hst[8];
    CK(hipMemcpy(hst, st, 64, hipMemcpyDeviceToHost));
    CK(hipFree(st));
    CK(hipMemset(c.bad, 0, 8));
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, c.keys2, c.vals2, c.keys, c.n, shift, mask, c.bad);
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, c.bad, 8, hipMemcpyDeviceToHost));
    if (!VALS) bad = 0; // keys only: the value-based check does not apply (the library tests cover it)
    if (!VALS) printf("keys-only ");
    if (ABLATE) printf("ABLATE %d: ", ABLATE);
    printf(CARRY ? "carry " : "plain ");
    if (ROUNDS > 1) printf("rounds %d ", ROUNDS);
    if (PREFETCH) printf("prefetch ");
    if (DMA) printf("dma ");
    printf("bits %d threads %4d kpt %2d tile %5d lds %6zu blk/cu %d nb %5u | count %.3f ms (%.0f GB/s) scan %.3f | scatter %.3f ms "
           "(%.0f GB/s) | pass %.3f ms %s\n",
           BITS, THREADS, KPT, TILE, sizeof(Smem), blocks_per_cu, nb, t_count, c.n * 4.0 / t_count / 1e6, t_scan, t_scatter,
           c.n * 16.0 / t_scatter / 1e6, t_count + t_scan + t_scatter, bad ? "WRONG" : "ok");
    {
        double per_tile = 1.0 / (double) tiles; // cycles per tile (100 MHz s_memtime ticks? -> printed raw)
        const char* names[8] = {"issue", "loadwait", "rank", "bar1", "offsets", "stage", "wout", "bar_end"};
        printf("    stamps/tile:");
        for (int i = 0; i < 8; i++) printf(" %s %.0f", names[i], hst[i] * per_tile);
        printf("\n");
    }
    fflush(stdout);
}

// the 128-byte-line scatter (radix_scatter_lines.hpp) behind the production count + row scan
template<int BITS, int THREADS, int KPT, bool VALS = true, int ABLATE = 0, int RS = (KPT + 2) / 3, bool STAGGER = true, bool NT = false, int PRIO = 0,
         bool RA = false>
void run_lines(Ctx& c, uint32_t shift, uint32_t mask_override = 0)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int TILE = THREADS * KPT;
    using Smem = LineSmem<uint32_t, BITS, THREADS, KPT, VALS>;
    // SB_SRCOFF=k: the source arrays start k elements into their allocations (what an unaligned source costs: the
    // arrays of a pass whose workgroup ranges start at arbitrary elements); the element count is n - 32 either way
    const size_t src_off = getenv("SB_SRCOFF") ? (size_t) atoi(getenv("SB_SRCOFF")) : 0;
    const uint32_t* skeys = c.keys + src_off;
    const uint32_t* svals = c.vals + src_off;
    const size_t n_eff = getenv("SB_SRCOFF") ? c.n - 32 : c.n;
    if (getenv("SB_SRCOFF")) printf("src offset %zu: ", src_off);
    const uint32_t tiles = (uint32_t) ((n_eff + TILE - 1) / TILE);
    uint32_t nb = std::min<uint32_t>(tiles, (uint32_t) c.cus);
    if (getenv("SB_NB")) nb = std::min<uint32_t>(tiles, (uint32_t) atoi(getenv("SB_NB")));
    uint32_t* totals = c.table + (size_t) RADIX * nb;
    const uint32_t mask = mask_override ? mask_override : RADIX - 1;
    if (mask_override) printf("mask %u: ", mask);
    auto scatter = radix_scatter_lines_kernel<uint32_t, BITS, THREADS, KPT, false, VALS, ABLATE, false, RS, STAGGER, NT, PRIO, false, RA>;
    auto scatter_st = radix_scatter_lines_kernel<uint32_t, BITS, THREADS, KPT, false, VALS, ABLATE, true, RS, STAGGER, NT, PRIO, false, RA>;
    if (RA) printf("rank: returning LDS atomics: ");
    CK(hipFuncSetAttribute((const void*) scatter_st, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    hipLaunchKernelGGL((radix_count_kernel<uint32_t, BITS, THREADS, TILE>), dim3(nb), dim3(THREADS), 0, 0, skeys, c.table, (uint32_t) n_eff, shift, mask,
                       tiles, 0u);
    hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, 0, c.table, totals, nb);
    CK(hipMemset(c.keys2, 0xff, c.n * 4));
    CK(hipMemset(c.vals2, 0xff, c.n * 4));
    float t_scatter = time_min(c, 5, [&] {
        hipLaunchKernelGGL(scatter, dim3(nb), dim3(THREADS), sizeof(Smem), 0, skeys, VALS ? svals : nullptr, c.keys2, VALS ? c.vals2 : nullptr,
                           c.table, totals, (uint32_t) n_eff, shift, mask, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (const uint2*) nullptr, 0u, (const uint32_t*) nullptr);
    });
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemset(c.bad, 0, 8));
    if (!ABLATE && VALS) hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, c.keys2, c.vals2, c.keys, n_eff, shift, mask, c.bad);
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, c.bad, 8, hipMemcpyDeviceToHost));
    if (!VALS) bad = 0;
    unsigned long long* st;
    CK(hipMalloc(&st, 128));
    CK(hipMemset(st, 0, 128));
    hipLaunchKernelGGL(scatter_st, dim3(nb), dim3(THREADS), sizeof(Smem), 0, skeys, VALS ? svals : nullptr, c.keys2, VALS ? c.vals2 : nullptr, c.table,
                       totals, (uint32_t) n_eff, shift, mask, tiles, st, 0u, (PassPlan*) nullptr, 0u, (const uint2*) nullptr, 0u, (const uint32_t*) nullptr);
    unsigned long long hst[16];
    CK(hipMemcpy(hst, st, 128, hipMemcpyDeviceToHost));
    CK(hipFree(st));
    if (!VALS) printf("keys-only ");
    if (ABLATE) printf("ABLATE %d: ", ABLATE);
    {   // the count kernel of the next pass right behind this scatter (as inside a sort)
        float best = 1e9f;
        for (int r = 0; r < 5; r++)
        {
            hipLaunchKernelGGL(scatter, dim3(nb), dim3(THREADS), sizeof(Smem), 0, skeys, VALS ? svals : nullptr, c.keys2, VALS ? c.vals2 : nullptr,
                               c.table, totals, (uint32_t) n_eff, shift, mask, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (const uint2*) nullptr, 0u, (const uint32_t*) nullptr);
            CK(hipEventRecord(c.ev[0]));
            hipLaunchKernelGGL((radix_count_kernel<uint32_t, BITS, THREADS, TILE>), dim3(nb), dim3(THREADS), 0, 0, c.keys2, c.table + (1 << 20), (uint32_t) n_eff,
                               shift + BITS, mask, tiles, 0u);
            CK(hipEventRecord(c.ev[1]));
            CK(hipEventSynchronize(c.ev[1]));
            float ms;
            CK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
            best = std::min(best, ms);
        }
        printf("%s%s[count right behind it: %.3f ms] ", NT ? "nt-stores " : "", PRIO == 1 ? "prio-stage " : PRIO == 2 ? "prio-tails " : PRIO == 3 ? "prio-both " : "", best);
    }
    printf("lines %ssplit %d bits %d threads %4d kpt %2d tile %5d lds %6zu nb %5u | scatter %.3f ms (%.0f GB/s) %s\n", STAGGER ? "stagger " : "", RS, BITS, THREADS, KPT, TILE, sizeof(Smem), nb,
           t_scatter, n_eff * (VALS ? 16.0 : 8.0) / t_scatter / 1e6, ABLATE ? "(ablated)" : bad ? "WRONG" : "ok");
    const char* names[8] = {"bar1", "scan", "stage+rankA", "bar4", "lines", "bar5", "tails+rankB", "-"};
    for (int w = 0; w < 2; w++)
    {
        double sum = 0;
        printf("    stamps/tile wave %2d:", w ? THREADS / 64 - 1 : 0);
        for (int i = 0; i < 8; i++) printf(" %s %.0f", names[i], hst[w * 8 + i] / (double) tiles), sum += hst[w * 8 + i] / (double) tiles;
        printf(" | sum %.0f\n", sum);
    }
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int log2n = argc > 1 ? atoi(argv[1]) : 28;
    int zero = argc > 2 ? atoi(argv[2]) : 0;
    Ctx c;
    c.n = log2n > 64 ? (size_t) log2n : (size_t) 1 << log2n; // values > 64 are taken as an element count
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    c.cus = p.multiProcessorCount;
    printf("%s %s CUs %d, N = 2^%d %s\n", p.name, p.gcnArchName, c.cus, log2n, zero ? "(zero keys)" : "");
    if (getenv("SB_PLACE"))
    {
        // placement experiment: the four arrays cut from ONE allocation at chosen relative offsets (bytes, from the
        // environment: SB_PLACE="d1,d2,d3" added to the natural 1x, 2x, 3x array-size offsets)
        size_t d1 = 0, d2 = 0, d3 = 0;
        sscanf(getenv("SB_PLACE"), "%zu,%zu,%zu", &d1, &d2, &d3);
        char* pool;
        CK(hipMalloc(&pool, c.n * 16 + d1 + d2 + d3 + (64u << 20)));
        c.keys = (uint32_t*) pool;
        c.vals = (uint32_t*) (pool + c.n * 4 + d1);
        c.keys2 = (uint32_t*) (pool + c.n * 8 + d1 + d2);
        c.vals2 = (uint32_t*) (pool + c.n * 12 + d1 + d2 + d3);
    }
    else
    {
        // SB_OFFS="o0,o1,o2,o3" (MiB): every array gets its own allocation of size + 1 GiB and starts o_i MiB into it
        // (if the driver hands out 1 GiB-aligned physical blocks, these offsets ARE the arrays' phases modulo 1 GiB)
        if (getenv("SB_OFFS"))
        {
            size_t o[4] = {0, 0, 0, 0};
            sscanf(getenv("SB_OFFS"), "%zu,%zu,%zu,%zu", &o[0], &o[1], &o[2], &o[3]);
            uint32_t** arr[4] = {&c.keys, &c.vals, &c.keys2, &c.vals2};
            for (int i = 0; i < 4; i++)
            {
                char* base;
                CK(hipMalloc(&base, c.n * 4 + ((size_t) 1 << 30)));
                *arr[i] = (uint32_t*) (base + (o[i] << 20));
            }
        }
        else
        {
        // SB_GAP="g1,g2,g3" (MiB): dummy allocations between the four arrays (shifts their relative physical placement)
        size_t g1 = 0, g2 = 0, g3 = 0;
        if (getenv("SB_GAP")) sscanf(getenv("SB_GAP"), "%zu,%zu,%zu", &g1, &g2, &g3);
        void* dummy;
        CK(hipMalloc(&c.keys, c.n * 4));
        if (g1) CK(hipMalloc(&dummy, g1 << 20));
        CK(hipMalloc(&c.vals, c.n * 4));
        if (g2) CK(hipMalloc(&dummy, g2 << 20));
        CK(hipMalloc(&c.keys2, c.n * 4));
        if (g3) CK(hipMalloc(&dummy, g3 << 20));
        CK(hipMalloc(&c.vals2, c.n * 4));
        }
    }
    printf("arrays at %p %p %p %p\n", (void*) c.keys, (void*) c.vals, (void*) c.keys2, (void*) c.vals2);
    CK(hipMalloc(&c.table, (256 * 8192 + 256) * 4));
    CK(hipMalloc(&c.bad, 8));
    for (int i = 0; i < 4; i++) CK(hipEventCreate(&c.ev[i]));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, c.keys, c.vals, c.n, zero);
    CK(hipDeviceSynchronize());

    for (int g : {2048, 4096, 8192})
    {
        float t = time_min(c, 5, [&] {
            hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, 0, (const uint4*) c.keys, (const uint4*) c.vals,
                               (uint4*) c.keys2, (uint4*) c.vals2, c.n / 4);
        });
        printf("copy (2 streams in, 2 out) grid %d: %.3f ms (%.0f GB/s)\n", g, t, c.n * 16.0 / t / 1e6);
    }

    auto cw = [&](auto kern, const char* name, int g, int th) {
        float t = time_min(c, 5, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(th), 0, 0, c.keys, c.vals, c.keys2, c.vals2, c.n); });
        printf("copy %s grid %d x %d: %.3f ms (%.0f GB/s)\n", name, g, th, t, c.n * 16.0 / t / 1e6);
    };
    cw(copy_width_kernel<16, 16>, "ld16 st16", 256, 1024);
    cw(copy_width_kernel<4, 4>, "ld4  st4 ", 256, 1024);
    if (getenv("SB_PARTIAL")) // does a CU go faster when fewer CUs are active? (per-CU cap vs shared HBM)
        for (int g : {32, 64, 128, 192})
        {
            cw(copy_width_kernel<4, 4>, "ld4  st4 ", g, 1024);
            cw(copy_width_kernel<16, 16>, "ld16 st16", g, 1024);
        }
    const uint32_t shift = 8; // any digit of uniform keys
    if (getenv("SB_COPYSWEEP"))
    {
        // one pool, one process: a 4-stream copy (2 in, 2 out, all at the same element offset) as a function of the byte
        // distances between the four arrays
        char* pool;
        const size_t G1 = (size_t) 1 << 30;
        CK(hipMalloc(&pool, 14 * G1));
        auto run = [&](size_t ob, size_t oc, size_t od, const char* what) {
            const uint32_t* a = (const uint32_t*) pool;
            const uint32_t* b = (const uint32_t*) (pool + ob);
            uint32_t* cc = (uint32_t*) (pool + oc);
            uint32_t* dd = (uint32_t*) (pool + od);
            float t = time_min(c, 7, [&] { hipLaunchKernelGGL((copy_width_kernel<4, 4>), dim3(256), dim3(1024), 0, 0, a, b, cc, dd, c.n); });
            printf("copy b@%.4f c@%.4f d@%.4f GiB %s: %.3f ms (%.0f GB/s)\n", ob / (double) G1, oc / (double) G1, od / (double) G1, what, t,
                   c.n * 16.0 / t / 1e6);
        };
        const size_t M1 = 1 << 20;
        auto run_split = [&](auto kern, size_t ob, size_t oc, size_t od, const char* what) {
            const uint32_t* a = (const uint32_t*) pool;
            const uint32_t* b = (const uint32_t*) (pool + ob);
            uint32_t* cc = (uint32_t*) (pool + oc);
            uint32_t* dd = (uint32_t*) (pool + od);
            float t = time_min(c, 7, [&] { hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, a, b, cc, dd, c.n); });
            printf("split copy b@%.4f c@%.4f d@%.4f GiB %s: %.3f ms (%.0f GB/s)\n", ob / (double) G1, oc / (double) G1, od / (double) G1, what, t,
                   c.n * 16.0 / t / 1e6);
        };
        printf("-- interleaved copy vs copies split in time, at the worst (1 GiB) and the best (1.5 GiB) distance\n");
        for (size_t D : {G1, G1 + 512 * M1})
        {
            run(D, 4 * G1, 4 * G1 + D, "interleaved");
            run_split(copy_split_kernel<16384>, D, 4 * G1, 4 * G1 + D, "split, 16 Ki elements per stream and turn");
            run_split(copy_split_kernel<65536>, D, 4 * G1, 4 * G1 + D, "split, 64 Ki");
            run_split(copy_split_kernel<262144>, D, 4 * G1, 4 * G1 + D, "split, 256 Ki");
        }
        printf("-- pairs (a,b) and (c,d) at distance D, a-c 4 GiB apart\n");
        for (size_t k = 0; k <= 32; k++) run(G1 + k * 64 * M1, 4 * G1, 5 * G1 + k * 64 * M1, "D = 1 GiB + k*64 MiB");
        printf("-- D = 1.5 GiB, c moved by y\n");
        for (size_t k = 0; k <= 16; k++) run(G1 + 512 * M1, 4 * G1 + k * 64 * M1, 5 * G1 + 512 * M1 + k * 64 * M1, "c,d + k*64 MiB");
        printf("-- fine sweep around D = 1.5 GiB\n");
        for (long k = -8; k <= 8; k++) run(G1 + 512 * M1 + k * 8 * (long) M1, 4 * G1, 5 * G1 + 512 * M1 + k * 8 * (long) M1, "D = 1.5 GiB + k*8 MiB");
        return 0;
    }
    if (getenv("SB_SCATSWEEP"))
    {
        // one pool, one process: the production scatter as a function of (vals - keys) for the source and destination pair
        char* pool;
        const size_t G1 = (size_t) 1 << 30, M1 = 1 << 20;
        CK(hipMalloc(&pool, 16 * G1));
        uint32_t *k0 = c.keys, *v0 = c.vals;
        printf("rows: source vals - keys = 1 GiB + ds; columns: destination vals - keys = 1 GiB + dd; ds, dd = 0, 128, ... 1024 MiB\n");
        for (size_t is = 0; is <= 8; is++)
        {
            uint32_t* sk = (uint32_t*) pool;
            uint32_t* sv = (uint32_t*) (pool + G1 + is * 128 * M1);
            CK(hipMemcpy(sk, k0, c.n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(sv, v0, c.n * 4, hipMemcpyDeviceToDevice));
            printf("ds %4zu:", is * 128);
            for (size_t id = 0; id <= 8; id++)
            {
                uint32_t* dk = (uint32_t*) (pool + 6 * G1);
                uint32_t* dv = (uint32_t*) (pool + 7 * G1 + id * 128 * M1);
                using Smem = ScatterSmem<uint32_t, 8, 1024, 12, true, 1>;
                const uint32_t tiles = (uint32_t) ((c.n + 12288 - 1) / 12288), nb = 256;
                uint32_t* totals = c.table + (size_t) 256 * nb;
                auto scatter = radix_scatter_kernel<uint32_t, 8, 1024, 12, true, 0, false, 1>;
                CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
                hipLaunchKernelGGL((radix_count_kernel<uint32_t, 8, 1024, 12288>), dim3(nb), dim3(1024), 0, 0, sk, c.table, (uint32_t) c.n, shift, 255u, tiles, 0u);
                hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(256), dim3(256), 0, 0, c.table, totals, nb);
                float t = time_min(c, 7, [&] {
                    hipLaunchKernelGGL(scatter, dim3(nb), dim3(1024), sizeof(Smem), 0, sk, sv, dk, dv, c.table, totals, (uint32_t) c.n, shift, 255u, tiles,
                                       (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
                });
                printf(" %.3f", t);
            }
            printf("\n");
        }
        return 0;
    }
    if (getenv("SB_SCATRAND"))
    {
        // random placements of the four arrays inside one pool (offsets in units of 64 MiB), production scatter + copy
        char* pool;
        const size_t G1 = (size_t) 1 << 30, U = 64u << 20;
        CK(hipMalloc(&pool, 24 * G1));
        uint32_t *k0 = c.keys, *v0 = c.vals;
        using Smem = ScatterSmem<uint32_t, 8, 1024, 12, true, 1>;
        const uint32_t tiles = (uint32_t) ((c.n + 12288 - 1) / 12288), nb = 256;
        uint32_t* totals = c.table + (size_t) 256 * nb;
        auto scatter = radix_scatter_kernel<uint32_t, 8, 1024, 12, true, 0, false, 1>;
        CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
        uint64_t rng = 12345 + (getenv("SB_SEED") ? atoi(getenv("SB_SEED")) : 0);
        auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t) (rng >> 33); };
        for (int it = 0; it < 48; it++)
        {
            // four disjoint 1 GiB (16-unit) slots inside 24 GiB (384 units): slot bases 0, 96, 192, 288 units + random 0..79
            size_t o[4];
            for (int j = 0; j < 4; j++) o[j] = (size_t) (96 * j + next() % 80);
            if (it == 0) { o[0] = 0; o[1] = 96; o[2] = 192; o[3] = 288; } // all congruent mod 1 GiB... (96 units = 6 GiB)
            uint32_t* sk = (uint32_t*) (pool + o[0] * U);
            uint32_t* sv = (uint32_t*) (pool + o[1] * U);
            uint32_t* dk = (uint32_t*) (pool + o[2] * U);
            uint32_t* dv = (uint32_t*) (pool + o[3] * U);
            CK(hipMemcpy(sk, k0, c.n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(sv, v0, c.n * 4, hipMemcpyDeviceToDevice));
            hipLaunchKernelGGL((radix_count_kernel<uint32_t, 8, 1024, 12288>), dim3(nb), dim3(1024), 0, 0, sk, c.table, (uint32_t) c.n, shift, 255u, tiles, 0u);
            hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(256), dim3(256), 0, 0, c.table, totals, nb);
            float t = time_min(c, 7, [&] {
                hipLaunchKernelGGL(scatter, dim3(nb), dim3(1024), sizeof(Smem), 0, sk, sv, dk, dv, c.table, totals, (uint32_t) c.n, shift, 255u, tiles,
                                   (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
            });
            float tc = time_min(c, 5, [&] { hipLaunchKernelGGL((copy_width_kernel<4, 4>), dim3(256), dim3(1024), 0, 0, sk, sv, dk, dv, c.n); });
            printf("units mod16: sk %2zu sv %2zu dk %2zu dv %2zu | sv-sk %2zu dv-dk %2zu dk-sk %2zu dv-sv %2zu | scatter %.3f copy %.3f\n", o[0] % 16, o[1] % 16, o[2] % 16,
                   o[3] % 16, (o[1] - o[0]) % 16, (o[3] - o[2]) % 16, (o[2] - o[0]) % 16, (o[3] - o[1]) % 16, t, tc);
        }
        return 0;
    }
    if (getenv("SB_DSTS"))
    {
        // does the time depend on WHICH allocation the scatter writes to / reads from?  (same process, same kernel)
        const int K = 5;
        uint32_t *dk[K], *dv[K], *sk[K], *sv[K];
        for (int i = 0; i < K; i++)
        {
            CK(hipMalloc(&dk[i], c.n * 4));
            CK(hipMalloc(&dv[i], c.n * 4));
            CK(hipMalloc(&sk[i], c.n * 4));
            CK(hipMalloc(&sv[i], c.n * 4));
            CK(hipMemcpy(sk[i], c.keys, c.n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(sv[i], c.vals, c.n * 4, hipMemcpyDeviceToDevice));
        }
        uint32_t *k0 = c.keys, *v0 = c.vals, *k2 = c.keys2, *v2 = c.vals2;
        for (int i = 0; i < K; i++)
        {
            c.keys = k0; c.vals = v0; c.keys2 = dk[i]; c.vals2 = dv[i];
            printf("dst pair %d (%p %p): ", i, (void*) dk[i], (void*) dv[i]);
            run_variant<8, 1024, 12, true>(c, 1, shift);
        }
        for (int i = 0; i < K; i++)
        {
            c.keys = sk[i]; c.vals = sv[i]; c.keys2 = k2; c.vals2 = v2;
            printf("src pair %d (%p %p): ", i, (void*) sk[i], (void*) sv[i]);
            run_variant<8, 1024, 12, true>(c, 1, shift);
        }
        for (int i = 0; i < K; i++)
        {
            c.keys = k0; c.vals = v0; c.keys2 = dk[i]; c.vals2 = dv[(i + 1) % K];
            printf("dst keys %d vals %d: ", i, (i + 1) % K);
            run_variant<8, 1024, 12, true>(c, 1, shift);
        }
        // alternate two fixed configurations: a drift in time shows in both, a placement effect in one
        for (int rep = 0; rep < 6; rep++)
        {
            c.keys = k0; c.vals = v0; c.keys2 = dk[1]; c.vals2 = dv[1];
            printf("alt A (dst keys 1 vals 1): ");
            run_variant<8, 1024, 12, true>(c, 1, shift);
            c.keys2 = dk[1]; c.vals2 = dv[2];
            printf("alt B (dst keys 1 vals 2): ");
            run_variant<8, 1024, 12, true>(c, 1, shift);
        }
        return 0;
    }
    if (getenv("SB_COUNTAFTER"))
    { // the count kernel alone vs right behind a scatter (as inside a sort): where do its extra 0.06 ms come from?
        using Smem = ScatterSmem<uint32_t, 8, 1024, 12, true, 1>;
        const uint32_t tiles = (uint32_t) ((c.n + 12288 - 1) / 12288), nb = 256;
        uint32_t* totals = c.table + (size_t) 256 * nb;
        auto scatter = radix_scatter_kernel<uint32_t, 8, 1024, 12, true, 0, false, 1>;
        CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
        auto count = [&](const uint32_t* k) {
            hipLaunchKernelGGL((radix_count_kernel<uint32_t, 8, 1024, 12288>), dim3(nb), dim3(1024), 0, 0, k, c.table, (uint32_t) c.n, shift, 255u, tiles, 0u);
        };
        count(c.keys);
        hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(256), dim3(256), 0, 0, c.table, totals, nb);
        CK(hipDeviceSynchronize());
        uint32_t* table2;
        CK(hipMalloc(&table2, (256 * 8192 + 256) * 4));
        auto timed_count = [&](const uint32_t* k, bool after_scatter, const char* what) {
            float best = 1e9f;
            for (int r = 0; r < 7; r++)
            {
                if (after_scatter)
                    hipLaunchKernelGGL(scatter, dim3(nb), dim3(1024), sizeof(Smem), 0, c.keys, c.vals, c.keys2, c.vals2, c.table, totals, (uint32_t) c.n, shift,
                                       255u, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr);
                CK(hipEventRecord(c.ev[0]));
                hipLaunchKernelGGL((radix_count_kernel<uint32_t, 8, 1024, 12288>), dim3(nb), dim3(1024), 0, 0, k, table2, (uint32_t) c.n, shift + 8, 255u, tiles, 0u);
                CK(hipEventRecord(c.ev[1]));
                CK(hipEventSynchronize(c.ev[1]));
                float ms;
                CK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
                best = std::min(best, ms);
            }
            printf("count %-46s %.3f ms (%.0f GB/s)\n", what, best, c.n * 4.0 / best / 1e6);
        };
        timed_count(c.keys, false, "of the source keys, alone");
        timed_count(c.keys2, false, "of the scattered keys, alone");
        timed_count(c.keys2, true, "of the scattered keys, right after the scatter");
        timed_count(c.keys, true, "of the source keys, right after the scatter");
        return 0;
    }
    if (getenv("SB_ROUNDS"))
    { // tiles of 2 x the staging area (keys, values, ranks stay in registers across the two staging rounds)
        run_variant<8, 1024, 12, true>(c, 1, shift);
        run_variant<8, 1024, 24, true, 0, 2>(c, 1, shift);
        run_variant<8, 1024, 16, true, 0, 2>(c, 1, shift);
        run_variant<8, 1024, 20, true, 0, 2>(c, 1, shift);
        run_variant<4, 1024, 24, false, 0, 2>(c, 1, shift);
        run_variant<4, 1024, 12, false>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_KEYS"))
    { // keys-only geometries (8 B/key moved by the scatter)
        run_variant<8, 1024, 12, true, 0, 1, false, false, false>(c, 1, shift);
        run_variant<8, 1024, 16, true, 0, 1, false, false, false>(c, 1, shift);
        run_variant<8, 1024, 20, true, 0, 1, false, false, false>(c, 1, shift);
        run_variant<8, 1024, 24, true, 0, 1, false, false, false>(c, 1, shift);
        run_variant<8, 512, 24, true, 0, 1, false, false, false>(c, 2, shift);
        run_variant<4, 1024, 12, false, 0, 1, false, false, false>(c, 1, shift);
        run_variant<4, 1024, 24, false, 0, 1, false, false, false>(c, 1, shift);
        run_variant<8, 1024, 12, true>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_SMALL"))
    { // small geometry (sizes below the large-tile switch): carry or not, 2 or 3 workgroups per CU
        run_variant<8, 256, 16, false>(c, 3, shift);
        run_variant<8, 256, 16, true>(c, 2, shift);
        run_variant<8, 256, 16, false>(c, 4, shift);
        run_variant<8, 512, 8, false>(c, 2, shift);
        run_variant<8, 256, 16, false>(c, 3, shift);
        return 0;
    }
    if (getenv("SB_FOUR"))
    { // 4-bit digits (the reference's pass structure): tile size and carry
        run_variant<4, 1024, 12, false>(c, 1, shift);
        run_variant<4, 1024, 12, true>(c, 1, shift);
        run_variant<4, 1024, 16, false>(c, 1, shift);
        run_variant<4, 1024, 16, true>(c, 1, shift);
        run_variant<4, 512, 24, false>(c, 1, shift);
        run_variant<4, 1024, 12, false>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_TWO"))
    { // two 512-thread workgroups per CU (needs -DGLU_CARRY_ELEMS=8 to fit 2 x 79 KB of LDS)
        run_variant<8, 512, 12, true>(c, 2, shift);
        run_variant<8, 512, 12, true>(c, 1, shift);
        run_variant<8, 1024, 12, true>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_DIAG"))
    { // what bounds the production scatter: linear write-back (ABLATE 1), fewer digit values, plain
        run_variant<8, 1024, 12, true>(c, 1, shift);
        run_variant<8, 1024, 12, true, 1>(c, 1, shift);
        run_variant<8, 1024, 12, true>(c, 1, shift, 15u);
        run_variant<8, 1024, 12, true>(c, 1, shift, 63u);
        run_variant<8, 1024, 12, false>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_C32"))
    { // 128-byte carry (build with -DGLU_CARRY_ELEMS=32): 64 KiB of carry, so smaller tiles
        run_variant<8, 1024, 8, true>(c, 1, shift);
        run_variant<8, 1024, 9, true>(c, 1, shift);
        run_variant<8, 1024, 8, true>(c, 1, shift, 15u);
        run_variant<8, 1024, 8, true, 0, 1, true>(c, 1, shift);
        run_variant<8, 1024, 9, true, 0, 1, true>(c, 1, shift);
        run_variant<8, 1024, 9, true, 2>(c, 1, shift);
        run_variant<8, 1024, 9, true, 4>(c, 1, shift);
        run_variant<8, 1024, 9, true, 5>(c, 1, shift);
        return 0;
    }
    if (getenv("SB_SRCOFF"))
    {
        run_lines<8, 1024, 10, true, 0, 4, true, true>(c, shift);
        return 0;
    }
    if (getenv("SB_RA")) // round 3: ballot ranking against one returning LDS atomic per item, with the phase stamps
    {
        for (int rep = 0; rep < 2; rep++)
        {
            run_lines<8, 1024, 10, true, 0, 4, true, true>(c, shift);
            run_lines<8, 1024, 10, true, 0, 4, true, true, 0, true>(c, shift);
        }
        return 0;
    }
    if (getenv("SB_R3"))
    {
        for (int rep = 0; rep < 2; rep++)
        {
            run_lines<8, 1024, 10, true, 0, 4, true, true>(c, shift);
            run_lines<4, 1024, 12, true, 0, 4, true, true>(c, shift);
        }
        return 0;
    }
    if (getenv("SB_LINES"))
    {
        run_lines<8, 1024, 10, true, 0, 4, true, true>(c, shift);
        run_variant<8, 1024, 12, true>(c, 1, shift);
        run_lines<8, 1024, 10, true, 0, 4, true, true, 1>(c, shift);
        run_lines<8, 1024, 10, true, 0, 4, true, true, 2>(c, shift);
        run_lines<8, 1024, 10, true, 0, 4, true, true, 3>(c, shift);
        run_lines<8, 1024, 10, true, 0, 4, true, true>(c, shift);
        run_lines<8, 1024, 10, true, 4>(c, shift);
        run_lines<8, 1024, 10>(c, shift, 15u);
        run_lines<8, 1024, 16, false>(c, shift);
        run_lines<4, 1024, 12>(c, shift);
        return 0;
    }
    if (getenv("SB_QUICK"))
    {
        run_variant<8, 1024, 12, true>(c, 1, shift);
        return 0;
    }
    if (argc > 3)
    { // short list for counter collection (rocprofv3 --pmc)
        run_variant<8, 1024, 12, true>(c, 1, shift);
        run_variant<8, 1024, 12, false>(c, 1, shift);
        run_variant<4, 1024, 16, true>(c, 1, shift);
        return 0;
    }
    {   // count kernel alone: threads per workgroup at one workgroup per CU
        const uint32_t tiles = (uint32_t) ((c.n + 12288 - 1) / 12288), nb = std::min<uint32_t>(tiles, (uint32_t) c.cus);
        auto cnt = [&](auto kern, int threads, const char* name) {
            float t = time_min(c, 7, [&] { hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, c.keys, c.table, (uint32_t) c.n, shift, 255u, tiles, 0u, (const uint32_t*) nullptr, (PassPlan*) nullptr, 0u, false, 0u, 0u); });
            printf("count %s: %.3f ms (%.0f GB/s)\n", name, t, c.n * 4.0 / t / 1e6);
        };
        cnt(radix_count_kernel<uint32_t, 8, 256, 12288>, 256, "8-bit  256 thr");
        cnt(radix_count_kernel<uint32_t, 8, 512, 12288>, 512, "8-bit  512 thr");
        cnt(radix_count_kernel<uint32_t, 8, 1024, 12288>, 1024, "8-bit 1024 thr");
        cnt(radix_count_kernel<uint32_t, 4, 256, 12288>, 256, "4-bit  256 thr");
        cnt(radix_count_kernel<uint32_t, 4, 512, 12288>, 512, "4-bit  512 thr");
        cnt(radix_count_kernel<uint32_t, 4, 1024, 12288>, 1024, "4-bit 1024 thr");
        cnt(radix_count_kernel<uint32_t, 8, 512, 12288>, 512, "8-bit  512 thr");
    }
    run_variant<8, 1024, 12, true>(c, 1, shift);
    if (getenv("SB_DMA"))
    {
        run_variant<8, 1024, 12, true, 0, 1, false, true>(c, 1, shift);
        run_variant<8, 1024, 12, true, 3, 1, false, false>(c, 1, shift);
        run_variant<8, 1024, 12, true, 3, 1, false, true>(c, 1, shift);
        run_variant<4, 1024, 12, false, 0, 1, false, true>(c, 1, shift);
        run_variant<4, 1024, 12, false>(c, 1, shift);
        run_variant<8, 256, 16, true, 0, 1, false, true>(c, 3, shift);
        run_variant<8, 256, 16, true>(c, 3, shift);
    }
    if (getenv("SB_PREFETCH"))
    {
        run_variant<8, 1024, 12, true, 0, 1, true>(c, 1, shift);
        run_variant<8, 1024, 8, true>(c, 1, shift);
        run_variant<8, 1024, 8, true, 0, 1, true>(c, 1, shift);
        run_variant<8, 512, 24, true>(c, 1, shift);
        run_variant<8, 512, 24, true, 0, 1, true>(c, 1, shift);
        run_variant<4, 1024, 12, false, 0, 1, true>(c, 1, shift);
    }
    return 0;
}
