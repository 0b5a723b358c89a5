// lds_final_pass_bench.hip -- round 4, pricing the last pass of the "sort that ends in LDS" (DESIGN.md section 8): after two
// most-significant-digit passes every run of `S` consecutive pairs shares its key's top 16 bits, and ONE pass can finish the
// sort by ordering each run by the low 16 bits inside LDS (two 8-bit rank / scan / re-stage rounds, like the library's
// single-workgroup sort), reading and writing each pair once: 16 B/pair instead of the 2 x 20.5 B/pair of two more passes.
// Whether that pays is a question of the pass's COMPUTE, which this harness measures on the best case (every run exactly one
// tile of S = THREADS x KPT pairs, so no ragged loads and no packing of runs into tiles): keys = (run index << 16) | random
// 16 bits, values = position.  Every geometry's output is compared with std::stable_sort on sampled tiles.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -o tools/lds_final_pass_bench tools/lds_final_pass_bench.hip
//   tools/lds_final_pass_bench [log2 pairs = 28]
// Second half: the pass in the product's form (runs of any length from an array of run starts) and what was tried on it --
// stores aligned to the run's first 128-byte line (slower: 1.05-1.09 against 1.02-1.06 ms), an equal share of the run per wave
// instead of 18 items whatever the length (slower alone, faster together with 16-bit counters, which bring the workgroup
// below 40 KiB and a fourth workgroup onto the CU: 0.96-0.99 ms, adopted), 6-byte stage slots for a fifth workgroup (level,
// measured in place only: its output check does not apply to this harness's keys), persistent workgroups that load the next
// run's bounds early (slower: 1.05-1.16 ms -- a workgroup per run lets the next one start while this one drains its stores).
// -DLFB_NT_LOADS: non-temporal loads of the run (level: 0.93 / 0.99 ms either way).
// Records: profiles/r04/lds_final_pass_*.txt.  Not part of the product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "radix_sort_kernels.hpp"

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

// One workgroup per tile (the hardware's dispatcher is the tile loop; two or more workgroups per CU overlap one tile's loads
// and stores with another's ranking).  The body follows radix_sort_single_block_kernel: wave-striped items, wave-private
// running digit counters, one scan over (digit, wave), staging in ranked order, reading back in wave-striped order.
template<int THREADS, int KPT, int DIGIT_BITS>
__global__ __launch_bounds__(THREADS) void lds_final_pass_kernel(const uint32_t* __restrict__ keys_in,
                                                                 const uint32_t* __restrict__ vals_in,
                                                                 uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                 uint32_t low_bits)
{
    using Smem = SingleBlockSmem<uint32_t, DIGIT_BITS, THREADS, KPT>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int WAVE_TILE = kWave * KPT;
    constexpr int WQ = WAVES / 4;
    constexpr int SCAN_THREADS = RADIX * WQ;
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    static_assert(WAVES % 4 == 0 && SCAN_THREADS <= THREADS, "offset scan geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wave_off = wave * WAVE_TILE + lane;
    const size_t base = (size_t) blockIdx.x * (THREADS * KPT);

    uint32_t key[KPT], val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        key[i] = keys_in[base + wave_off + i * kWave];
        val[i] = vals_in[base + wave_off + i * kWave];
    }

    uint32_t* my_cnt = s.wcnt[wave];
    for (uint32_t shift = 0; shift < low_bits; shift += DIGIT_BITS)
    {
        constexpr uint32_t MASK = (1u << DIGIT_BITS) - 1;
        for (int i = tid; i < WAVES * Smem::WCNT_STRIDE; i += THREADS) (&s.wcnt[0][0])[i] = 0;
        __syncthreads();

        uint32_t rank[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t d = (key[i] >> shift) & MASK;
            uint32_t* const cnt = my_cnt + d;
            const uint32_t prev = *cnt;
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < DIGIT_BITS; bit++)
            {
                int32_t sel;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = (uint32_t) __popc(plo) + (uint32_t) __popc(phi);
            rank[i] = prev + lower;
            asm volatile("" : "+v"(rank[i]));
            *cnt = prev + total;
        }
        __syncthreads();

        {
            const uint32_t sd = tid / WQ, sw = (tid % WQ) * 4;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (tid < SCAN_THREADS)
            {
                c0 = s.wcnt[sw + 0][sd];
                c1 = s.wcnt[sw + 1][sd];
                c2 = s.wcnt[sw + 2][sd];
                c3 = s.wcnt[sw + 3][sd];
            }
            uint32_t excl = 0;
            if (wave < SCAN_WAVES)
            {
                uint32_t wtotal;
                excl = wave_exclusive_sum(c0 + c1 + c2 + c3, lane, wtotal);
                if (SCAN_WAVES > 1 && lane == 0) s.scan_tmp[wave] = wtotal;
            }
            if (SCAN_WAVES > 1)
            {
                __syncthreads();
                excl += sum_of_preceding_waves(s.scan_tmp, SCAN_WAVES, wave, lane);
            }
            if (tid < SCAN_THREADS)
            {
                s.wcnt[sw + 0][sd] = excl;
                s.wcnt[sw + 1][sd] = excl + c0;
                s.wcnt[sw + 2][sd] = excl + c0 + c1;
                s.wcnt[sw + 3][sd] = excl + c0 + c1 + c2;
            }
        }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < KPT; i++) s.stage.put(my_cnt[(key[i] >> shift) & MASK] + rank[i], key[i], val[i]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; i++) s.stage.get(wave_off + i * kWave, key[i], val[i]);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        __builtin_nontemporal_store(key[i], &keys_out[base + wave_off + i * kWave]);
        __builtin_nontemporal_store(val[i], &vals_out[base + wave_off + i * kWave]);
    }
}

// The product's form of the pass (radix_lds_finish.hpp): runs of any length from `starts`.  in / out may be the same arrays
// (the product sorts in place).  DYN: every wave takes an equal share of the run (a multiple of 64 slots) and ranks only the
// items its share has, instead of KPT items whatever the run's length.  CntT: type of the wave-private counters (16-bit
// ones bring 256 x 18 down to 38 KiB: four workgroups per CU instead of three).  ALIGN_STORES: the final stores start at the
// 128-byte line the run starts in.
template<typename CntT, int WAVES, int RADIX>
struct RunsSmemCnt
{
    CntT wcnt[WAVES][RADIX];
};
// PACK: the stage holds the low 16 key bits (the top 16 are the run's) and the value, 6 instead of 8 bytes per slot.
template<int THREADS, int KPT, bool ALIGN_STORES, bool DYN, typename CntT, bool PACK = false>
__global__ __launch_bounds__(THREADS) void lds_runs_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                                           uint32_t* vals_out, const uint32_t* __restrict__ starts, uint32_t num_runs)
{
    constexpr int RADIX = 256;
    constexpr int WAVES = THREADS / kWave;
    constexpr int TILE = THREADS * KPT;
    constexpr int WQ = WAVES / 4;
    constexpr int SCAN_THREADS = RADIX * WQ;
    constexpr int SCAN_WAVES = (SCAN_THREADS + kWave - 1) / kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr size_t SLOT = PACK ? 6 : 8;
    uint2* stage = reinterpret_cast<uint2*>(smem_raw);
    uint32_t* stage_v = reinterpret_cast<uint32_t*>(smem_raw);
    uint16_t* stage_k = reinterpret_cast<uint16_t*>(smem_raw + (size_t) TILE * 4);
    CntT(*wcnt)[RADIX] = reinterpret_cast<CntT(*)[RADIX]>(smem_raw + (size_t) TILE * SLOT);
    uint32_t* scan_tmp = reinterpret_cast<uint32_t*>(smem_raw + (size_t) TILE * SLOT + sizeof(CntT) * WAVES * RADIX);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (a launch with fewer workgroups than runs: every workgroup takes runs blockIdx.x, + gridDim.x, ...; the next run's
    // bounds are loaded while this one is sorted)
    uint32_t nbegin = starts[blockIdx.x], nend = starts[blockIdx.x + 1];
    for (uint32_t run = blockIdx.x; run < num_runs; run += gridDim.x)
    {
    const uint32_t begin = nbegin, len = nend - nbegin;
    if (run + gridDim.x < num_runs) nbegin = starts[run + gridDim.x], nend = starts[run + gridDim.x + 1];
    const uint32_t chunk = DYN ? ((len + WAVES * 64 - 1) / (WAVES * 64)) * 64 : (uint32_t) (kWave * KPT);
    const uint32_t items = chunk >> 6;
    const uint32_t wave_off = wave * chunk + lane;
    uint32_t key[KPT], val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        const uint32_t p = wave_off + i * kWave;
        const bool ok = p < len;
        const uint32_t pc = ok ? p : len - 1; // unconditional loads of an element of the run; slots past its end become pads
#ifdef LFB_NT_LOADS
        const uint32_t k = __builtin_nontemporal_load(&keys_in[begin + pc]), v = __builtin_nontemporal_load(&vals_in[begin + pc]);
#else
        const uint32_t k = keys_in[begin + pc], v = vals_in[begin + pc];
#endif
        key[i] = ok ? k : ~0u;
        val[i] = ok ? v : 0u;
    }
    CntT* my_cnt = wcnt[wave];
    for (uint32_t shift = 0; shift < 16; shift += 8)
    {
        for (int i = tid; i < WAVES * RADIX * (int) sizeof(CntT) / 4; i += THREADS) reinterpret_cast<uint32_t*>(&wcnt[0][0])[i] = 0;
        __syncthreads();
        uint32_t rank[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if (DYN && i >= items) continue;
            const uint32_t d = (key[i] >> shift) & 255u;
            CntT* const cnt = my_cnt + d;
            const uint32_t prev = *cnt;
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < 8; bit++)
            {
                int32_t sel;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = (uint32_t) __popc(plo) + (uint32_t) __popc(phi);
            rank[i] = prev + lower;
            asm volatile("" : "+v"(rank[i]));
            *cnt = (CntT) (prev + total);
        }
        __syncthreads();
        {
            const uint32_t sd = tid / WQ, sw = (tid % WQ) * 4;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (tid < SCAN_THREADS)
            {
                c0 = wcnt[sw + 0][sd];
                c1 = wcnt[sw + 1][sd];
                c2 = wcnt[sw + 2][sd];
                c3 = wcnt[sw + 3][sd];
            }
            uint32_t excl = 0;
            if (wave < SCAN_WAVES)
            {
                uint32_t wtotal;
                excl = wave_exclusive_sum(c0 + c1 + c2 + c3, lane, wtotal);
                if (SCAN_WAVES > 1 && lane == 0) scan_tmp[wave] = wtotal;
            }
            if (SCAN_WAVES > 1)
            {
                __syncthreads();
                excl += sum_of_preceding_waves(scan_tmp, SCAN_WAVES, wave, lane);
            }
            if (tid < SCAN_THREADS)
            {
                wcnt[sw + 0][sd] = (CntT) excl;
                wcnt[sw + 1][sd] = (CntT) (excl + c0);
                wcnt[sw + 2][sd] = (CntT) (excl + c0 + c1);
                wcnt[sw + 3][sd] = (CntT) (excl + c0 + c1 + c2);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            if (DYN && i >= items) continue;
            const uint32_t pos = (uint32_t) my_cnt[(key[i] >> shift) & 255u] + rank[i];
            if (PACK)
                stage_k[pos] = (uint16_t) key[i], stage_v[pos] = val[i];
            else
                stage[pos] = make_uint2(key[i], val[i]);
        }
        __syncthreads();
        if (shift == 0 || !ALIGN_STORES)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                if (DYN && i >= items) continue;
                if (PACK)
                    key[i] = (key[i] & 0xFFFF0000u) | stage_k[wave_off + i * kWave], val[i] = stage_v[wave_off + i * kWave];
                else
                {
                    const uint2 e = stage[wave_off + i * kWave];
                    key[i] = e.x, val[i] = e.y;
                }
            }
            if (shift == 0) __syncthreads();
        }
    }
    if (ALIGN_STORES)
    {
        const uint32_t lead = begin & 31u;
        const uint32_t woff = wave * (kWave * KPT) + lane;
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t q = woff + i * kWave;
            if (q >= lead && q - lead < len)
            {
                const uint2 e = stage[q - lead];
                __builtin_nontemporal_store(e.x, &keys_out[begin - lead + q]);
                __builtin_nontemporal_store(e.y, &vals_out[begin - lead + q]);
            }
        }
    }
    else
    {
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t p = wave_off + i * kWave;
            if ((!DYN || i < items) && p < len)
            {
                __builtin_nontemporal_store(key[i], &keys_out[begin + p]);
                __builtin_nontemporal_store(val[i], &vals_out[begin + p]);
            }
        }
    }
    __syncthreads(); // (the stage and the counters are reused)
    }
}

// the same traffic with no sorting work: what the memory system gives this access pattern
template<int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void copy_tiles_kernel(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wave_off = wave * (kWave * KPT) + lane;
    const size_t base = (size_t) blockIdx.x * (THREADS * KPT);
    uint32_t key[KPT], val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        key[i] = keys_in[base + wave_off + i * kWave];
        val[i] = vals_in[base + wave_off + i * kWave];
    }
#pragma unroll
    for (int i = 0; i < KPT; i++)
    {
        __builtin_nontemporal_store(key[i], &keys_out[base + wave_off + i * kWave]);
        __builtin_nontemporal_store(val[i], &vals_out[base + wave_off + i * kWave]);
    }
}

struct Buffers
{
    uint32_t *ki, *vi, *ko, *vo;
    size_t n;
    std::vector<uint32_t> host_keys;
};

template<int THREADS, int KPT, int DIGIT_BITS>
static void run(const char* name, Buffers& b, bool copy_only = false)
{
    constexpr uint32_t S = THREADS * KPT;
    using Smem = SingleBlockSmem<uint32_t, DIGIT_BITS, THREADS, KPT>;
    const uint32_t tiles = (uint32_t) (b.n / S);
    auto kern = lds_final_pass_kernel<THREADS, KPT, DIGIT_BITS>;
    CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 12; rep++)
    {
        CK(hipEventRecord(e0, 0));
        if (copy_only)
            hipLaunchKernelGGL((copy_tiles_kernel<THREADS, KPT>), dim3(tiles), dim3(THREADS), 0, 0, b.ki, b.vi, b.ko, b.vo);
        else
            hipLaunchKernelGGL(kern, dim3(tiles), dim3(THREADS), sizeof(Smem), 0, b.ki, b.vi, b.ko, b.vo, 16u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    // check sampled tiles against std::stable_sort on the low 16 bits
    size_t bad = 0;
    if (!copy_only)
    {
        std::vector<uint32_t> ok(S), ov(S);
        for (uint32_t t : {0u, 1u, tiles / 3, tiles / 2 + 7, tiles - 1})
        {
            if (t >= tiles) continue;
            CK(hipMemcpy(ok.data(), b.ko + (size_t) t * S, S * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ov.data(), b.vo + (size_t) t * S, S * 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> idx(S);
            for (uint32_t i = 0; i < S; i++) idx[i] = i;
            const uint32_t* hk = b.host_keys.data() + (size_t) t * S;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return (hk[x] & 0xffffu) < (hk[y] & 0xffffu); });
            for (uint32_t i = 0; i < S; i++)
                if (ok[i] != hk[idx[i]] || ov[i] != (uint32_t) ((size_t) t * S + idx[i])) bad++;
        }
    }
    printf("%-34s tile %5u pairs  LDS %6zu B  median %.3f ms  min %.3f ms  %.2f TB/s (16 B/pair)  %s\n", name, S,
           copy_only ? (size_t) 0 : sizeof(Smem), med, ms.front(), 16.0 * b.n / (med * 1e-3) / 1e12,
           copy_only ? "" : (bad ? "MISMATCH" : "matches std::stable_sort on 5 tiles"));
    if (bad) exit(1);
}


template<int THREADS, int KPT, bool ALIGN_STORES, bool DYN, typename CntT, bool PACK = false>
static void run_runs(const char* name, Buffers& b, const uint32_t* d_starts, const std::vector<uint32_t>& h_starts, bool in_place,
                     uint32_t grid = 0)
{
    const uint32_t runs = (uint32_t) h_starts.size() - 1;
    const size_t lds = (size_t) THREADS * KPT * (PACK ? 6 : 8) + sizeof(CntT) * (THREADS / 64) * 256 + 64;
    auto kern = lds_runs_kernel<THREADS, KPT, ALIGN_STORES, DYN, CntT, PACK>;
    CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    if (in_place) CK(hipMemcpy(b.ki, b.host_keys.data(), b.n * 4, hipMemcpyHostToDevice)); // (sorted runs sort as fast as random ones)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 12; rep++)
    {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid ? grid : runs), dim3(THREADS), lds, 0, b.ki, b.vi, in_place ? b.ki : b.ko, in_place ? b.vi : b.vo, d_starts, runs);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    size_t bad = 0;
    if (!in_place)
    {
        for (uint32_t t : {0u, 1u, runs / 3, runs / 2 + 7, runs - 1})
        {
            const uint32_t lo = h_starts[t], S = h_starts[t + 1] - lo;
            std::vector<uint32_t> ok(S), ov(S), idx(S);
            CK(hipMemcpy(ok.data(), b.ko + lo, S * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ov.data(), b.vo + lo, S * 4, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < S; i++) idx[i] = i;
            const uint32_t* hk = b.host_keys.data() + lo;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return (hk[x] & 0xffffu) < (hk[y] & 0xffffu); });
            for (uint32_t i = 0; i < S; i++)
                if ((ok[i] & 0xFFFFu) != (hk[idx[i]] & 0xFFFFu) || ov[i] != lo + idx[i]) bad++; // (PACK: the top bits are the run's first key's)
        }
    }
    printf("%-72s LDS %6zu B  median %.3f ms  min %.3f ms  %s\n", name, lds, ms[ms.size() / 2], ms.front(),
           in_place ? "" : (bad ? "MISMATCH" : "ok"));
    if (bad) exit(1);
}

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    Buffers b;
    b.n = (size_t) 1 << lg;
    CK(hipMalloc(&b.ki, b.n * 4));
    CK(hipMalloc(&b.vi, b.n * 4));
    CK(hipMalloc(&b.ko, b.n * 4));
    CK(hipMalloc(&b.vo, b.n * 4));
    b.host_keys.resize(b.n);
    std::vector<uint32_t> hv(b.n);
    std::mt19937 rng(1234);
    for (size_t i = 0; i < b.n; i++)
    {
        b.host_keys[i] = (uint32_t) ((i >> 10) << 16) | (rng() & 0xffffu); // top bits: finer than any tile, never decreasing
        hv[i] = (uint32_t) i;
    }
    CK(hipMemcpy(b.ki, b.host_keys.data(), b.n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.vi, hv.data(), b.n * 4, hipMemcpyHostToDevice));
    printf("last pass of a sort that ends in LDS, 2^%d pairs (u32 key + u32 value), every run exactly one tile\n", lg);
    run<1024, 4, 8>("copy only 1024 x 4", b, true);
    run<1024, 4, 8>("1024 threads x 4, 2 x 8-bit rounds", b);
    run<1024, 5, 8>("1024 threads x 5, 2 x 8-bit rounds", b);
    run<512, 8, 8>("512 threads x 8, 2 x 8-bit rounds", b);
    run<512, 4, 8>("512 threads x 4, 2 x 8-bit rounds", b);
    run<256, 16, 8>("256 threads x 16, 2 x 8-bit rounds", b);
    run<256, 8, 8>("256 threads x 8, 2 x 8-bit rounds", b);
    run<1024, 8, 8>("1024 threads x 8, 2 x 8-bit rounds", b);
    // the product's form: 65536 runs; (a) all of 4096 pairs, (b) lengths 4096 +- up to 300 (what uniform keys give)
    {
        const uint32_t runs = (uint32_t) (b.n / 4096);
        std::vector<uint32_t> even(runs + 1), ragged(runs + 1);
        for (uint32_t r = 0; r <= runs; r++) even[r] = r * 4096u;
        ragged[0] = 0;
        for (uint32_t r = 1; r < runs; r++) ragged[r] = r * 4096u + (rng() % 601u) - 300u;
        ragged[runs] = (uint32_t) b.n;
        uint32_t *d_even, *d_ragged;
        CK(hipMalloc(&d_even, (runs + 1) * 4));
        CK(hipMalloc(&d_ragged, (runs + 1) * 4));
        CK(hipMemcpy(d_even, even.data(), (runs + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_ragged, ragged.data(), (runs + 1) * 4, hipMemcpyHostToDevice));
        run_runs<256, 16, false, false, uint32_t>("runs of 4096, 256 x 16", b, d_even, even, false);
        run_runs<256, 18, false, false, uint32_t>("runs of 4096, 256 x 18", b, d_even, even, false);
        run_runs<256, 18, false, true, uint32_t>("runs of 4096, 256 x 18, equal shares per wave", b, d_even, even, false);
        run_runs<256, 18, false, true, uint16_t>("runs of 4096, 256 x 18, equal shares per wave, 16-bit counters", b, d_even, even, false);
        run_runs<256, 18, true, false, uint32_t>("ragged runs, 256 x 18, aligned stores", b, d_ragged, ragged, false);
        run_runs<256, 18, false, false, uint32_t>("ragged runs, 256 x 18", b, d_ragged, ragged, false);
        run_runs<256, 18, false, true, uint32_t>("ragged runs, 256 x 18, equal shares per wave", b, d_ragged, ragged, false);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares per wave, 16-bit counters", b, d_ragged, ragged, false);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares per wave, 16-bit counters, in place", b, d_ragged, ragged, true);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares, 16-bit counters, 1024 persistent workgroups", b, d_ragged, ragged, false, 1024);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares, 16-bit counters, 2048 persistent workgroups", b, d_ragged, ragged, false, 2048);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares, 16-bit counters, 8192 workgroups x 8 runs", b, d_ragged, ragged, false, 8192);
        run_runs<256, 18, false, true, uint16_t>("ragged runs, 256 x 18, equal shares, 16-bit counters, 1024 persistent workgroups, in place", b, d_ragged, ragged, true, 1024);
        run_runs<256, 18, false, true, uint16_t, true>("ragged runs, 256 x 18, equal shares, 16-bit counters, 6-byte slots (time only)", b, d_ragged, ragged, true);
        run_runs<512, 9, false, false, uint32_t>("ragged runs, 512 x 9", b, d_ragged, ragged, false);
        run_runs<512, 9, false, true, uint32_t>("ragged runs, 512 x 9, equal shares per wave", b, d_ragged, ragged, false);
        run_runs<512, 9, false, true, uint16_t>("ragged runs, 512 x 9, equal shares per wave, 16-bit counters", b, d_ragged, ragged, false);
        run_runs<1024, 5, false, true, uint16_t>("ragged runs, 1024 x 5, equal shares per wave, 16-bit counters", b, d_ragged, ragged, false);
    }
    return 0;
}
