// interleaved_pattern_bench.hip -- round 4, second question after packed_pattern_bench.hip: keep keys and values in
// separate LINES (32 elements per 128-byte line each, as the line scatter writes them today) but put the sorter's two
// scratch arrays into ONE allocation as alternating blocks: 2^S keys, then their 2^S values, then the next 2^S keys ...
//   element i:  key at word ((i >> S) << (S + 1)) + (i & (2^S - 1)),  value 2^S words behind it
// The key line and the value line of a run then lie a FIXED 4 * 2^S bytes apart inside the same physical page (pages are
// at least 2^21 bytes), so whether the two streams collide in the memory system no longer depends on where two separate
// allocations happened to fall -- if some block size is reliably in the fast class, placement by measurement
// (tune_scratch_placement) can go.  Forms, all with whole 128-byte line stores, non-temporal, 256 destination regions:
//   2->2   separate arrays both sides (today)          2->I  caller arrays -> interleaved scratch
//   I->2   interleaved scratch -> caller arrays        sum   2 x (2->I + I->2) against 2 x (2->2 + 2->2 back)
// timed on several placements of the arrays like packed_pattern_bench.hip.  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/interleaved_pattern_bench tools/interleaved_pattern_bench.hip
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

// word index of key i in an interleaved array with blocks of 2^S elements
__device__ inline size_t ikey(size_t i, uint32_t S) { return ((i >> S) << (S + 1)) + (i & (((size_t) 1 << S) - 1)); }

// src_s / dst_s: 0 = separate arrays (ka, va / kb, vb), otherwise the block shift S of the interleaved array (ia / ib)
template<int RUN>
__global__ __launch_bounds__(1024) void line_pattern_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ va,
                                                            const uint32_t* __restrict__ ia, uint32_t* __restrict__ kb,
                                                            uint32_t* __restrict__ vb, uint32_t* __restrict__ ib, uint32_t per_wg,
                                                            uint32_t region_len, uint32_t src_s, uint32_t dst_s)
{
    constexpr uint32_t VTILE = 256u * RUN, STEP = 1024u * 4u;
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
            if (src_s)
            {
                const size_t w = ikey(e, src_s);
                r[h][0] = *reinterpret_cast<const u32x4*>(ia + w);
                r[h][1] = *reinterpret_cast<const u32x4*>(ia + w + ((size_t) 1 << src_s));
            }
            else
            {
                r[h][0] = *reinterpret_cast<const u32x4*>(ka + e);
                r[h][1] = *reinterpret_cast<const u32x4*>(va + e);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const uint32_t x = x0 + h * STEP + tid * 4u;
            const uint32_t vt = x / VTILE, w = x - vt * VTILE;
            const uint32_t c = w / RUN, o = w - c * RUN;
            const uint32_t j = (c + b * 37u + vt * 11u) & 255u;
            const size_t d = (size_t) j * region_len + ((size_t) b * vtiles + vt) * RUN + o;
            if (dst_s)
            {
                const size_t wd = ikey(d, dst_s);
                store_nt(reinterpret_cast<u32x4*>(ib + wd), r[h][0]);
                store_nt(reinterpret_cast<u32x4*>(ib + wd + ((size_t) 1 << dst_s)), r[h][1]);
            }
            else
            {
                store_nt(reinterpret_cast<u32x4*>(kb + d), r[h][0]);
                store_nt(reinterpret_cast<u32x4*>(vb + d), r[h][1]);
            }
        }
    }
}

struct Stat
{
    std::vector<float> ms;
    void add(float x) { ms.push_back(x); }
    void print(const char* name, double bytes) const
    {
        std::vector<float> s = ms;
        std::sort(s.begin(), s.end());
        const float med = s[s.size() / 2];
        printf("%-44s median %.3f ms (%.0f GB/s)  min %.3f  max %.3f  |", name, med, bytes / med / 1e6, s.front(), s.back());
        for (float x : ms) printf(" %.3f", x);
        printf("\n");
    }
};

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const int placements = argc > 2 ? atoi(argv[2]) : 8;
    const size_t n = (size_t) 1 << log2n, slack = 1 << 20;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256;
    const uint32_t per_wg = (uint32_t) (n / wgs), region_len = (uint32_t) (n / 256);
    const std::vector<uint32_t> shifts = {5, 8, 10, 12, 14, 16, 18, 19, 20, 22};
    Stat s22, s22back, ssum_sep;
    std::vector<Stat> s2i(shifts.size()), si2(shifts.size()), ssum(shifts.size());
    for (int p = 0; p < placements; p++)
    {
        std::vector<void*> spacers;
        auto spacer = [&](size_t bytes) {
            if (!bytes) return;
            void* s;
            CK(hipMalloc(&s, bytes));
            spacers.push_back(s);
        };
        uint32_t *ka, *va, *kb, *vb, *il;
        const size_t gap = (size_t) p * (512u << 20) / 2 + (p ? (96u << 20) : 0);
        CK(hipMalloc(&ka, n * 4 + slack));
        spacer(gap);
        CK(hipMalloc(&va, n * 4 + slack));
        spacer(gap / 2);
        CK(hipMalloc(&kb, n * 4 + slack));
        spacer(gap);
        CK(hipMalloc(&vb, n * 4 + slack));
        spacer(gap / 3);
        CK(hipMalloc(&il, n * 8 + (64u << 20)));
        CK(hipMemset(ka, 1, n * 4));
        CK(hipMemset(va, 2, n * 4));
        CK(hipMemset(il, 3, n * 8));
        auto time_it = [&](auto launch) {
            float best = 1e9f;
            for (int r = 0; r < 4; r++)
            {
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) best = std::min(best, ms);
            }
            return best;
        };
#define LP(KA, VA, IA, KB, VB, IB, SS, DS) \
    hipLaunchKernelGGL((line_pattern_kernel<32>), dim3(wgs), dim3(1024), 0, 0, KA, VA, IA, KB, VB, IB, per_wg, region_len, SS, DS)
        const float f22 = time_it([&] { LP(ka, va, nullptr, kb, vb, nullptr, 0u, 0u); });
        const float f22b = time_it([&] { LP(kb, vb, nullptr, ka, va, nullptr, 0u, 0u); });
        s22.add(f22), s22back.add(f22b), ssum_sep.add(2 * (f22 + f22b));
        for (size_t k = 0; k < shifts.size(); k++)
        {
            const uint32_t S = shifts[k];
            const float a = time_it([&] { LP(ka, va, nullptr, nullptr, nullptr, il, 0u, S); });
            const float b = time_it([&] { LP(nullptr, nullptr, il, ka, va, nullptr, S, 0u); });
            s2i[k].add(a), si2[k].add(b), ssum[k].add(2 * (a + b));
        }
        for (void* q : {(void*) ka, (void*) va, (void*) kb, (void*) vb, (void*) il}) CK(hipFree(q));
        for (void* s : spacers) CK(hipFree(s));
    }
    printf("2^%d pairs, %d placements; 16 B/pair per launch\n", log2n, placements);
    const double bytes = n * 16.0;
    s22.print("2->2 caller -> separate scratch", bytes);
    s22back.print("2->2 separate scratch -> caller", bytes);
    ssum_sep.print("4 passes, separate scratch (sum)", 4 * bytes);
    for (size_t k = 0; k < shifts.size(); k++)
    {
        char name[96];
        snprintf(name, sizeof name, "2->I blocks of 2^%u elements (%u KiB)", shifts[k], (4u << shifts[k]) >> 10);
        s2i[k].print(name, bytes);
        snprintf(name, sizeof name, "I->2 blocks of 2^%u", shifts[k]);
        si2[k].print(name, bytes);
        snprintf(name, sizeof name, "4 passes, interleaved 2^%u (sum)", shifts[k]);
        ssum[k].print(name, 4 * bytes);
    }
    return 0;
}
