// rank_microbench.hip -- what the phases of the line scatter kernel cost in isolation (tuning harness, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -o tools/rank_microbench tools/rank_microbench.hip
// One 1024-thread workgroup per CU (16 waves, 4 per SIMD: the geometry of radix_scatter_lines_kernel) runs REPS rounds of one
// phase on register / LDS data only (no global memory in the loop) and reports shader cycles per round (s_memtime of wave 0
// and wave 15) for:
//   valu     N dependent-free v_add_u32 per lane: the vector issue rate with 4 waves per SIMD
//   rank     the match-any ranking of KPT items (8 ballots + 16 v_bitop3 + mbcnt/bcnt + counter read/write per item)
//   ranknolds  the same without the LDS counter
//   stage    KPT ds_write_b64 to random positions (the staging writes: bank conflicts as in the kernel)
//   stageseq the same to consecutive positions (conflict-free)
#include <hip/hip_runtime.h>

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

constexpr int KPT = 10;
constexpr int THREADS = 1024;
constexpr int TILE = THREADS * KPT;

template<int MODE>
__global__ __launch_bounds__(THREADS) void phase_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         unsigned long long* __restrict__ cycles, int reps)
{
    __shared__ uint16_t wcnt[16][260];
    __shared__ uint2 buf[TILE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t key[KPT], rank[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++) key[i] = in[(blockIdx.x * TILE + i * THREADS + tid) & 0xFFFFF], rank[i] = 0;
    for (int i = tid; i < 16 * 260 / 2; i += THREADS) reinterpret_cast<uint32_t*>(&wcnt[0][0])[i] = 0;
    __syncthreads();
    uint16_t* const my_cnt = wcnt[wave];
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++)
    {
        if (MODE == 0)
        {
            // 400 independent-ish VALU adds per round (4 chains)
            uint32_t a = key[0], b = key[1], c = key[2], d = key[3];
#pragma unroll
            for (int i = 0; i < 100; i++)
            {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(r));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(r));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(r));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(r));
            }
            acc += a + b + c + d;
        }
        if (MODE == 1 || MODE == 2)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t d = __builtin_amdgcn_ubfe(key[i] + r, 8, 8);
                uint16_t* const cnt = my_cnt + d;
                const uint32_t prev = MODE == 1 ? *cnt : rank[i];
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
                rank[i] = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, prev));
                asm volatile("" : "+v"(rank[i]));
                uint32_t new_count;
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(new_count) : "v"(plo), "v"(prev));
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(new_count) : "v"(phi), "v"(new_count));
                if (MODE == 1) *cnt = (uint16_t) new_count;
                else acc += new_count;
            }
#pragma unroll
            for (int i = 0; i < KPT; i++) acc += rank[i];
            if (MODE == 1)
                for (int i = lane; i < 130; i += 64) reinterpret_cast<uint32_t*>(my_cnt)[i] = 0;
        }
        if (MODE == 3 || MODE == 4)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                uint32_t pos;
                if (MODE == 3)
                {
                    // a random permutation-like position inside the tile (what a ranked position looks like to the banks)
                    uint32_t x = (key[i] + r * 0x9E3779B9u) * 0x85EBCA6Bu;
                    pos = (x >> 8) % TILE;
                }
                else
                    pos = (i * THREADS + tid + r) % TILE;
                buf[pos] = make_uint2(key[i], r);
            }
            acc += buf[(tid * 7 + r) % TILE].x;
        }
        if (MODE == 15 || MODE == 16)
        {
            // does a wave64 instruction whose EXEC mask covers one 32-lane half only issue in half the time?
            uint32_t a = key[0], b = key[1], c = key[2], d = key[3];
            if (MODE == 15 ? lane < 32 : (lane & 1) == 0)
            {
#pragma unroll
                for (int i = 0; i < 100; i++)
                {
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(a) : "v"(c), "v"(b));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(c) : "v"(a), "v"(d));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(b) : "v"(d), "v"(a));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(d) : "v"(b), "v"(c));
                }
            }
            acc += a + b + c + d;
        }
        if (MODE >= 6 && MODE <= 14)
        {
            uint32_t a = key[0], b = key[1], c = key[2], d = key[3];
            unsigned long long m0 = 0, m1 = 0;
#pragma unroll
            for (int i = 0; i < 100; i++)
            {
                if (MODE == 6) // VALU compare writing an SGPR pair
                {
                    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m0) : "v"(a));
                    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m1) : "v"(b));
                    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m0) : "v"(c));
                    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m1) : "v"(d));
                }
                if (MODE == 7) // v_bitop3 with an SGPR operand (the mask was written long ago)
                {
                    const uint32_t s0 = __builtin_amdgcn_readfirstlane(r);
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(a) : "s"(s0), "v"(b));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(c) : "s"(s0), "v"(d));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(b) : "s"(s0), "v"(a));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(d) : "s"(s0), "v"(c));
                }
                if (MODE == 8)
                {
                    asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(a) : "v"(b));
                    asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(c) : "v"(d));
                    asm volatile("v_bfe_i32 %0, %1, 2, 1" : "=v"(a) : "v"(b));
                    asm volatile("v_bfe_i32 %0, %1, 2, 1" : "=v"(c) : "v"(d));
                }
                if (MODE == 9) // v_bitop3, all operands in vector registers
                {
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(a) : "v"(c), "v"(b));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(c) : "v"(a), "v"(d));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(b) : "v"(d), "v"(a));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(d) : "v"(b), "v"(c));
                }
                if (MODE == 11) // v_bfe_i32 with offset and width in vector registers (are the inline constants the cost?)
                {
                    uint32_t three = key[4] | 3u, one = key[5] | 1u;
                    asm volatile("v_bfe_i32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(three), "v"(one));
                    asm volatile("v_bfe_i32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(three), "v"(one));
                    asm volatile("v_bfe_i32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(one), "v"(one));
                    asm volatile("v_bfe_i32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(one), "v"(one));
                }
                if (MODE == 12) // compare writing VCC (e32 encoding)
                {
                    asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0" : : "v"(a) : "vcc");
                    asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0" : : "v"(b) : "vcc");
                    asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0" : : "v"(c) : "vcc");
                    asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0" : : "v"(d) : "vcc");
                }
                if (MODE == 13) // shifts / adds with an inline constant (VOP2): is any constant slow, or only VOP3 / bfe?
                {
                    asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
                    asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(b) : "v"(a));
                    asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(c));
                    asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(d) : "v"(c));
                }
                if (MODE == 14) // v_and_b32 with an SGPR operand (VOP2)
                {
                    const uint32_t s0 = __builtin_amdgcn_readfirstlane(r);
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(b) : "s"(s0));
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(c) : "s"(s0));
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(d) : "s"(s0));
                }
                if (MODE == 10) // compare -> SGPR, consumed right away by two v_bitop3 (the ranking's pattern), 4 per group
                {
                    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m0) : "v"(a));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(b) : "s"((uint32_t) m0), "v"(a));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(c) : "s"((uint32_t) (m0 >> 32)), "v"(a));
                    asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(a) : "v"(d));
                }
            }
            acc += a + b + c + d + (uint32_t) m0 + (uint32_t) m1;
        }
        if (MODE == 5)
        {
            // the staging loop as in the kernel: digit -> prefix read (LDS u16) -> position -> ds_write_b64
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t d = __builtin_amdgcn_ubfe(key[i] + r, 8, 8);
                uint32_t x = (key[i] + r * 0x9E3779B9u) * 0x85EBCA6Bu;
                const uint32_t pos = ((x >> 8) + my_cnt[d]) % TILE;
                buf[pos] = make_uint2(key[i], r);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (wave == 0 || wave == 15)) atomicAdd(&cycles[wave ? 1 : 0], t1 - t0);
    out[blockIdx.x * THREADS + tid] = acc;
}

template<int MODE>
void run(const char* name, const uint32_t* in, uint32_t* out, unsigned long long* cyc, int blocks, int reps, double per_round_unit,
         const char* unit)
{
    CK(hipMemset(cyc, 0, 16));
    hipLaunchKernelGGL(phase_kernel<MODE>, dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, 4);
    CK(hipDeviceSynchronize());
    CK(hipMemset(cyc, 0, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(phase_kernel<MODE>, dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    const double c0 = (double) h[0] / blocks / reps, c15 = (double) h[1] / blocks / reps;
    printf("%-10s %8.0f cycles/round (wave 0)  %8.0f (wave 15)  = %.2f %s   [%.3f ms, %.2f GHz]\n", name, c0, c15, c15 / per_round_unit, unit, ms,
           c15 * reps / (ms * 1e6));
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount;
    uint32_t *in, *out;
    unsigned long long* cyc;
    CK(hipMalloc(&in, 4 << 20));
    CK(hipMalloc(&out, (size_t) blocks * THREADS * 4));
    CK(hipMalloc(&cyc, 16));
    std::vector<uint32_t> h(1 << 20);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h)
    {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        v = (uint32_t) (x >> 16);
    }
    CK(hipMemcpy(in, h.data(), 4 << 20, hipMemcpyHostToDevice));
    const int reps = 2000;
    // per SIMD: 4 waves x instructions per wave
    run<0>("valu", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per wave64 VALU instruction and SIMD");
    run<1>("rank", in, out, cyc, blocks, reps, KPT, "cycles per item (4 waves per SIMD ranking together)");
    run<2>("ranknolds", in, out, cyc, blocks, reps, KPT, "cycles per item");
    run<3>("stage", in, out, cyc, blocks, reps, 16.0 * KPT, "cycles per wave-level ds_write_b64 (random positions)");
    run<4>("stageseq", in, out, cyc, blocks, reps, 16.0 * KPT, "cycles per wave-level ds_write_b64 (consecutive positions)");
    run<5>("stage+cnt", in, out, cyc, blocks, reps, 16.0 * KPT, "cycles per wave-level (u16 read + ds_write_b64)");
    run<6>("cmp->sgpr", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_cmp_e64 writing an SGPR pair");
    run<7>("bitop3 s", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bitop3 with an SGPR operand");
    run<8>("bfe_i32", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bfe_i32");
    run<9>("bitop3 v", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bitop3, vector operands");
    run<11>("bfe vgpr", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bfe_i32 with vector-register offset / width");
    run<12>("cmp->vcc", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_cmp_e32 writing VCC");
    run<13>("shift imm", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per VOP2 shift with an inline constant");
    run<14>("and sgpr", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_and_b32 with an SGPR operand");
    run<15>("half exec", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bitop3 (vector operands) with EXEC = lanes 0-31 only");
    run<16>("even lanes", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per v_bitop3 (vector operands) with EXEC = even lanes only");
    run<10>("cmp+2bitop", in, out, cyc, blocks, reps, 4 * 400.0, "cycles per instruction of (cmp -> sgpr, bitop3, bitop3, bfe)");
    return 0;
}
