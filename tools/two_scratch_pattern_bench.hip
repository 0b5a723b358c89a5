// two_scratch_pattern_bench.hip -- round 4, third layout question: a sort's four passes all touch the CALLER's pair of arrays
// (caller -> scratch -> caller -> scratch -> caller), and a pass runs in the fast class only if both its pairs fell well
// (packed_pattern_bench.hip).  The sorter can choose its scratch pair by measurement but not the caller's.  With TWO scratch
// pairs A and B, both chosen by measurement, only the first pass reads and only the last pass writes the caller's arrays:
//     caller -> A -> B -> A -> caller        against        caller -> A -> caller -> A -> caller
// This harness measures both chains on the line scatter's access pattern (no sorting work, whole 128-byte line stores, 256
// regions): `CANDS` candidate buffers are allocated behind spacers, the fastest pair (A) for caller -> pair and then the
// fastest pair (B) for A -> pair among the rest are found by timing, and both chains are timed on `CALLERS` caller pairs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/two_scratch_pattern_bench tools/two_scratch_pattern_bench.hip
// Not part of the product.
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
    const int CANDS = argc > 2 ? atoi(argv[2]) : 8, CALLERS = argc > 3 ? atoi(argv[3]) : 6;
    const size_t n = (size_t) 1 << log2n, slack = 1 << 20;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256;
    const uint32_t per_wg = (uint32_t) (n / wgs), region_len = (uint32_t) (n / 256);
    auto alloc = [&](size_t spacer_bytes) {
        void* sp = nullptr;
        if (spacer_bytes) CK(hipMalloc(&sp, spacer_bytes));
        u32x4* p;
        CK(hipMalloc(&p, n * 4 + slack));
        CK(hipMemset(p, 1, n * 4));
        if (sp) CK(hipFree(sp)); // the array stays where it is
        return p;
    };
    auto time_it = [&](const u32x4* ka, const u32x4* va, u32x4* kb, u32x4* vb) {
        float best = 1e9f;
        for (int r = 0; r < 3; r++)
        {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(line_pattern_kernel, dim3(wgs), dim3(1024), 0, 0, ka, va, kb, vb, per_wg, region_len);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = std::min(best, ms);
        }
        return best;
    };
    // the caller pairs: plain consecutive allocations, as an application makes them
    std::vector<std::pair<u32x4*, u32x4*>> callers;
    for (int c = 0; c < CALLERS; c++)
    {
        u32x4* k = alloc(0);
        u32x4* v = alloc((size_t) (c % 3) * (160u << 20));
        callers.push_back({k, v});
    }
    // scratch candidates behind spacers of 0, 0.5 ... GiB
    std::vector<u32x4*> cand;
    for (int i = 0; i < CANDS; i++) cand.push_back(alloc((size_t) i * (512u << 20) / 2));
    // A: the fastest (key, value) choice among the candidates as DESTINATION of caller 0; B: the fastest among the rest as
    // destination of A (what a prepare-time search would do with its own calibration arrays)
    int a0 = 0, a1 = 1, b0 = -1, b1 = -1;
    float best = 1e9f;
    for (int i = 0; i < CANDS; i++)
        for (int j = 0; j < CANDS; j++)
            if (i != j)
            {
                const float t = time_it(callers[0].first, callers[0].second, cand[i], cand[j]);
                if (t < best) best = t, a0 = i, a1 = j;
            }
    printf("A = candidates (%d, %d): caller 0 -> A %.3f ms\n", a0, a1, best);
    best = 1e9f;
    for (int i = 0; i < CANDS; i++)
        for (int j = 0; j < CANDS; j++)
            if (i != j && i != a0 && i != a1 && j != a0 && j != a1)
            {
                const float t = time_it(cand[a0], cand[a1], cand[i], cand[j]) + time_it(cand[i], cand[j], cand[a0], cand[a1]);
                if (t < best) best = t, b0 = i, b1 = j;
            }
    printf("B = candidates (%d, %d): A -> B + B -> A %.3f ms\n", b0, b1, best);
    u32x4 *Ak = cand[a0], *Av = cand[a1], *Bk = cand[b0], *Bv = cand[b1];
    printf("%-8s %9s %9s %9s %9s | %12s %12s\n", "caller", "c -> A", "A -> c", "A -> B", "B -> A", "c-A-c-A-c", "c-A-B-A-c");
    double sum_one = 0, sum_two = 0;
    for (int c = 0; c < CALLERS; c++)
    {
        const float ca = time_it(callers[c].first, callers[c].second, Ak, Av);
        const float ac = time_it(Ak, Av, callers[c].first, callers[c].second);
        const float ab = time_it(Ak, Av, Bk, Bv), ba = time_it(Bk, Bv, Ak, Av);
        const float one = 2 * (ca + ac), two = ca + ab + ba + ac;
        sum_one += one, sum_two += two;
        printf("%-8d %9.3f %9.3f %9.3f %9.3f | %12.3f %12.3f\n", c, ca, ac, ab, ba, one, two);
    }
    printf("mean over %d caller pairs: one scratch pair %.3f ms, two scratch pairs %.3f ms (%.1f %%)\n", CALLERS, sum_one / CALLERS,
           sum_two / CALLERS, (sum_two / sum_one - 1) * 100);
    return 0;
}
