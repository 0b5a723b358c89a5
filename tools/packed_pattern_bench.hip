// packed_pattern_bench.hip -- round 4: what would a PACKED (key, val) scratch array buy the 128-byte-line scatter?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/packed_pattern_bench tools/packed_pattern_bench.hip
// The access pattern of a counting pass with 256 digit values and no sorting work at all (like pattern_bench.hip), every
// store a whole 128-byte line written by 8 lanes x 16 bytes, non-temporal, in four forms:
//   2->2  keys[] + vals[]  ->  keys[] + vals[]      what every pass of the library does today (4 concurrent streams)
//   2->1  keys[] + vals[]  ->  pairs[]              a pass that WRITES a packed scratch (passes 0 and 2 of a 4-pass sort)
//   1->2  pairs[]          ->  keys[] + vals[]      a pass that READS it (passes 1 and 3)
//   1->1  pairs[]          ->  pairs[]              for reference
// A run of RUN elements of one (tile, digit) is RUN/32 lines of keys + RUN/32 lines of values, or RUN/16 lines of pairs:
// the same number of lines either way.  Every form is timed on several PLACEMENTS of its arrays (fresh allocations behind
// spacers of p x 512 MiB + 96 MiB), because the placement of the arrays moves the 4-stream form by 5-8 %:
// the question is whether fewer streams on one side move the median AND the minimum.  Not part of the product.
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

// Workgroup b owns elements [b * per_wg, (b + 1) * per_wg); its range is cut into virtual tiles of 256 runs of RUN
// elements; run c of virtual tile vt goes to region j(c, b, vt) at element offset j * region_len + (b * vtiles + vt) * RUN.
// A thread moves 4 elements (32 bytes) per step; 8 consecutive lanes move 32 consecutive elements.
template<int RUN, bool SRC_PACKED, bool DST_PACKED>
__global__ __launch_bounds__(1024) void line_pattern_kernel(const u32x4* __restrict__ ka, const u32x4* __restrict__ va,
                                                            const u32x4* __restrict__ pa, u32x4* __restrict__ kb,
                                                            u32x4* __restrict__ vb, u32x4* __restrict__ pb, uint32_t per_wg,
                                                            uint32_t region_len)
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
            const size_t e = base + x0 + h * STEP + tid * 4u; // first of this thread's 4 elements
            if (SRC_PACKED)
            {
                // 32 elements = two lines of pairs; lane g of the 8 reads 16 bytes of each line (whole-line loads)
                const size_t grp = e & ~(size_t) 31, g = (e >> 2) & 7;
                r[h][0] = pa[grp / 2 + g];     // pairs grp + 2g, 2g + 1
                r[h][1] = pa[grp / 2 + 8 + g]; // pairs grp + 16 + 2g, 2g + 1
            }
            else
            {
                r[h][0] = ka[e / 4];
                r[h][1] = va[e / 4];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const uint32_t x = x0 + h * STEP + tid * 4u;
            const uint32_t vt = x / VTILE, w = x - vt * VTILE;
            const uint32_t c = w / RUN, o = w - c * RUN;
            const uint32_t j = (c + b * 37u + vt * 11u) & 255u;
            const size_t d = (size_t) j * region_len + ((size_t) b * vtiles + vt) * RUN + o; // element index, multiple of 4
            if (DST_PACKED)
            {
                const size_t grp = d & ~(size_t) 31, g = (d >> 2) & 7;
                store_nt(&pb[grp / 2 + g], r[h][0]);
                store_nt(&pb[grp / 2 + 8 + g], r[h][1]);
            }
            else
            {
                store_nt(&kb[d / 4], r[h][0]);
                store_nt(&vb[d / 4], r[h][1]);
            }
        }
    }
}

// plain streaming copy of the same bytes (the ceiling of the device and placement)
template<bool SRC_PACKED, bool DST_PACKED>
__global__ __launch_bounds__(1024) void line_copy_kernel(const u32x4* __restrict__ ka, const u32x4* __restrict__ va,
                                                         const u32x4* __restrict__ pa, u32x4* __restrict__ kb, u32x4* __restrict__ vb,
                                                         u32x4* __restrict__ pb, uint32_t per_wg)
{
    const size_t base = (size_t) blockIdx.x * per_wg / 4;
    for (uint32_t x = threadIdx.x; x < per_wg / 4; x += 2048)
    {
        u32x4 a0, a1, b0, b1;
        if (SRC_PACKED)
        {
            a0 = pa[2 * (base + x)], a1 = pa[2 * (base + x) + 1];
            b0 = pa[2 * (base + x + 1024)], b1 = pa[2 * (base + x + 1024) + 1];
        }
        else
        {
            a0 = ka[base + x], a1 = va[base + x];
            b0 = ka[base + x + 1024], b1 = va[base + x + 1024];
        }
        if (DST_PACKED)
        {
            store_nt(&pb[2 * (base + x)], a0), store_nt(&pb[2 * (base + x) + 1], a1);
            store_nt(&pb[2 * (base + x + 1024)], b0), store_nt(&pb[2 * (base + x + 1024) + 1], b1);
        }
        else
        {
            store_nt(&kb[base + x], a0), store_nt(&vb[base + x], a1);
            store_nt(&kb[base + x + 1024], b0), store_nt(&vb[base + x + 1024], b1);
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
        printf("%-34s median %.3f ms (%.0f GB/s)  min %.3f (%.0f)  max %.3f (%.0f)  |", name, med, bytes / med / 1e6, s.front(),
               bytes / s.front() / 1e6, s.back(), bytes / s.back() / 1e6);
        for (float x : ms) printf(" %.3f", x);
        printf("\n");
    }
};

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const int placements = argc > 2 ? atoi(argv[2]) : 10;
    const size_t n = (size_t) 1 << log2n;
    const size_t slack = 1 << 20;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256;
    const uint32_t per_wg = (uint32_t) (n / wgs);
    const uint32_t region_len = (uint32_t) (n / 256);
    enum { F22, F21, F12, F11, C22, C21, C12, F22L, F21L, F12L, SORT4_SEP, SORT4_PACKED, SORT4_ALL_PACKED, NFORMS };
    const char* names[NFORMS] = {"2->2 runs of 32 (1+1 lines)", "2->1 runs of 32 (2 lines)", "1->2 runs of 32", "1->1 runs of 32",
                                 "copy 2->2", "copy 2->1", "copy 1->2", "2->2 runs of 64 (2+2 lines)", "2->1 runs of 64 (4 lines)",
                                 "1->2 runs of 64", "4 passes, separate scratch (sum)", "4 passes, packed scratch (sum)",
                                 "2->1, 1->1, 1->1, 1->2 (sum)"};
    Stat st[NFORMS];
    for (int p = 0; p < placements; p++)
    {
        // caller arrays (ka, va), separate scratch (kb, vb), packed scratch (pp): fresh allocations, spacers between them
        std::vector<void*> spacers;
        auto spacer = [&](size_t bytes) {
            if (!bytes) return;
            void* s;
            CK(hipMalloc(&s, bytes));
            spacers.push_back(s);
        };
        u32x4 *ka, *va, *kb, *vb, *pp, *pq;
        const size_t gap = (size_t) p * (512u << 20) / 2 + (p ? (96u << 20) : 0);
        CK(hipMalloc(&ka, n * 4 + slack));
        spacer(gap);
        CK(hipMalloc(&va, n * 4 + slack));
        spacer(gap / 2);
        CK(hipMalloc(&kb, n * 4 + slack));
        spacer(gap);
        CK(hipMalloc(&vb, n * 4 + slack));
        spacer(gap / 3);
        CK(hipMalloc(&pp, n * 8 + slack));
        spacer(gap / 5);
        CK(hipMalloc(&pq, n * 8 + slack)); // a second packed array: passes 1 and 2 of a fully packed pipeline (1->1)
        CK(hipMemset(ka, 1, n * 4));
        CK(hipMemset(va, 2, n * 4));
        CK(hipMemset(pp, 3, n * 8));
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
#define LP(RUN, S, D, KA, VA, PA, KB, VB, PB) \
    hipLaunchKernelGGL((line_pattern_kernel<RUN, S, D>), dim3(wgs), dim3(1024), 0, 0, KA, VA, PA, KB, VB, PB, per_wg, region_len)
        const float f22 = time_it([&] { LP(32, false, false, ka, va, nullptr, kb, vb, nullptr); });
        const float f22b = time_it([&] { LP(32, false, false, kb, vb, nullptr, ka, va, nullptr); });
        const float f21 = time_it([&] { LP(32, false, true, ka, va, nullptr, nullptr, nullptr, pp); });
        const float f12 = time_it([&] { LP(32, true, false, nullptr, nullptr, pp, ka, va, nullptr); });
        const float f11 = time_it([&] { LP(32, true, true, nullptr, nullptr, pp, nullptr, nullptr, pq); });
        const float f11b = time_it([&] { LP(32, true, true, nullptr, nullptr, pq, nullptr, nullptr, pp); });
        st[F22].add(f22), st[F21].add(f21), st[F12].add(f12), st[F11].add(f11);
        st[SORT4_SEP].add(2 * (f22 + f22b)), st[SORT4_PACKED].add(2 * (f21 + f12));
        st[SORT4_ALL_PACKED].add(f21 + f11 + f11b + f12);
        st[F22L].add(time_it([&] { LP(64, false, false, ka, va, nullptr, kb, vb, nullptr); }));
        st[F21L].add(time_it([&] { LP(64, false, true, ka, va, nullptr, nullptr, nullptr, pp); }));
        st[F12L].add(time_it([&] { LP(64, true, false, nullptr, nullptr, pp, ka, va, nullptr); }));
#define LC(S, D, KA, VA, PA, KB, VB, PB) \
    hipLaunchKernelGGL((line_copy_kernel<S, D>), dim3(wgs), dim3(1024), 0, 0, KA, VA, PA, KB, VB, PB, per_wg)
        st[C22].add(time_it([&] { LC(false, false, ka, va, nullptr, kb, vb, nullptr); }));
        st[C21].add(time_it([&] { LC(false, true, ka, va, nullptr, nullptr, nullptr, pp); }));
        st[C12].add(time_it([&] { LC(true, false, nullptr, nullptr, pp, ka, va, nullptr); }));
        for (void* q : {(void*) ka, (void*) va, (void*) kb, (void*) vb, (void*) pp, (void*) pq}) CK(hipFree(q));
        for (void* s : spacers) CK(hipFree(s));
    }
    printf("2^%d pairs, %d placements; 16 B/pair per launch\n", log2n, placements);
    for (int f = 0; f < NFORMS; f++) st[f].print(names[f], (f >= SORT4_SEP ? 4.0 : 1.0) * n * 16.0);
    return 0;
}
