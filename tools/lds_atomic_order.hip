// lds_atomic_order.hip -- EXPERIMENT, not part of the product: what would ranking with returning LDS atomics cost, and in
// which order does the LDS serve lanes of one wave that add to the same address?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/lds_atomic_order tools/lds_atomic_order.hip
// The scatter kernel ranks with 8 ballots per item (radix_scatter_lines.hpp: 170 cycles per item and wave, 44 % of its compute).
// `r = ds_add_rtn(&count[digit], 1)` would hand every lane a distinct slot in one instruction -- but a stable pass needs the
// slots of equal digits in LANE ORDER, and the ISA documents no order for same-address lanes of one instruction.  This program
// (a) times the instruction on wave-private counters with 10 items per lane, like the kernel's tile, and (b) counts, over many
// random digit vectors with heavy duplication, the wave-instructions whose same-digit lanes did not receive increasing values.
// An observed order is an observation, not a contract: the library does not use it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int THREADS = 1024, WAVES = 16, KPT = 10, RADIX = 256;

template<bool CHECK>
__global__ __launch_bounds__(THREADS) void rank_atomic_kernel(const uint32_t* __restrict__ digits, uint32_t* __restrict__ ranks,
                                                              unsigned long long* __restrict__ cycles, unsigned long long* violations, int rounds)
{
    __shared__ uint32_t cnt[WAVES][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t d[KPT];
    const uint32_t* src = digits + ((size_t) blockIdx.x * THREADS + tid) * KPT;
    for (int i = 0; i < KPT; i++) d[i] = src[i];
    uint32_t r[KPT];
    unsigned long long bad = 0, t_sum = 0;
    for (int round = 0; round < rounds; round++)
    {
        for (int i = lane; i < RADIX; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < KPT; i++) r[i] = atomicAdd(&cnt[wave][d[i]], 1u);
        __builtin_amdgcn_s_waitcnt(0);
        t_sum += __builtin_readcyclecounter() - t0;
        // order check: for every item, the lanes below me with my digit must all have received smaller values, and exactly
        // (my value - value of the lowest such lane ... ) -- simplest complete test: my value == (count of the digit before this
        // instruction) + (number of lower lanes with my digit).  The count before = minimum over the group.
#pragma unroll
        for (int i = 0; CHECK && i < KPT; i++)
        {
            // number of lower lanes with the same digit (ballot match over the 8 bits)
            uint64_t peers = ~0ull;
            for (int b = 0; b < 8; b++)
            {
                const uint64_t m = __ballot((d[i] >> b) & 1u);
                peers &= ((d[i] >> b) & 1u) ? m : ~m;
            }
            const uint32_t lower = __popcll(peers & ((1ull << lane) - 1ull));
            const int first = __ffsll((unsigned long long) peers) - 1;
            const uint32_t base = __shfl((int) r[i], first); // value the group's first lane received
            if (r[i] != base + lower) bad++;
        }
        for (int i = 0; i < KPT; i++) d[i] = (d[i] * 13u + r[i] + round) & (blockIdx.x & 1 ? 255u : 7u); // new digits: half of the workgroups draw from 8 values
    }
    for (int i = 0; i < KPT; i++) ranks[((size_t) blockIdx.x * THREADS + tid) * KPT + i] = r[i];
    if (lane == 0) cycles[blockIdx.x * WAVES + wave] = t_sum;
    if (bad) atomicAdd(violations, bad);
}

int main()
{
    const int blocks = 256, rounds = 2000;
    const size_t n = (size_t) blocks * THREADS * KPT;
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (uint32_t) (rand() & 255);
    uint32_t *dd, *dr;
    unsigned long long *dc, *dv;
    CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&dr, n * 4)); CK(hipMalloc(&dc, blocks * WAVES * 8)); CK(hipMalloc(&dv, 8));
    CK(hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dv, 0, 8));
    hipLaunchKernelGGL(rank_atomic_kernel<true>, dim3(blocks), dim3(THREADS), 0, 0, dd, dr, dc, dv, rounds);
    CK(hipDeviceSynchronize());
    // timing: no check, wall clock of the whole kernel (one workgroup per CU, 4 waves per SIMD), per digit range
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    for (int pass = 0; pass < 2; pass++)
    {
        // all workgroups with the same digit range: even block ids draw from 8 values, odd ones from 256 -> launch only one parity
        // (grid of 256 workgroups, parity chosen by an offset of the digit array is not needed: the range depends on blockIdx & 1,
        // so run 512 workgroups and time them together is not clean; instead run the kernel twice with all-even / all-odd ids
        // emulated through the rounds' first rewrite: simply time both and report the mix)
        const int r2 = 20000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rank_atomic_kernel<false>, dim3(blocks), dim3(THREADS), 0, 0, dd, dr, dc, dv, r2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = (double) ms * 1e-3 * clock_khz * 1e3;
        printf("  wall clock: %.3f ms for %d rounds x %d items (mixed digit ranges, incl. the 4 counter-zeroing writes and the digit update per round): %.1f cycles per item and SIMD (4 waves) at %.2f GHz nominal\n",
               ms, r2, KPT, cyc / ((double) r2 * KPT), clock_khz * 1e-6);
    }
    std::vector<unsigned long long> c(blocks * WAVES);
    unsigned long long v = 0;
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&v, dv, 8, hipMemcpyDeviceToHost));
    double wide = 0, narrow = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < WAVES; w++) (b & 1 ? wide : narrow) += (double) c[b * WAVES + w];
    const double per = (double) rounds * KPT * (blocks / 2) * WAVES;
    printf("returning LDS atomic add, wave-private counters, %d items per lane, 16 waves per CU:\n", KPT);
    printf("  digits from 256 values: %.1f cycles per item and wave (s_memtime clock)\n", wide / per);
    printf("  digits from   8 values: %.1f cycles per item and wave\n", narrow / per);
    printf("  items whose value was not (value of the group's first lane + number of lower lanes with the same digit): %llu of %.0f\n", v,
           (double) rounds * KPT * blocks * THREADS);
    return 0;
}
