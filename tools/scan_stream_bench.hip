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

namespace glu_hip
{
// ---- EXPERIMENT (measured, not adopted: profiles/r04/scan_chained_variants.txt): the chained scan as a STREAM -- persistent
// workgroups, the next chunk in flight during the look-back.  It was in the library for one commit (7ae0df2).
// scan_chunks_kernel<CHAINED> runs one workgroup per chunk: ticket (a returning global atomic), loads, local scan, look-back,
// stores -- and between the arrival of its loads and the first of its stores a workgroup has nothing in flight (the
// look-back is one or two cross-CU round trips, 3-5 us of a chunk's ~26 us), nor between its last store and the first load
// of the workgroup dispatched after it.  2^28 uint32: 0.42-0.44 ms where the same bytes with no carry between workgroups
// take 0.372 ms (scan_small_partitions_kernel).  Here a workgroup stays and takes chunk after chunk in ticket order, two
// tickets ahead: the loads of chunk i + 1 are issued BEFORE the local scan and the look-back of chunk i, the ticket of chunk
// i + 2 is requested at the same time and collected after the look-back.  Tickets keep the order free of deadlocks whatever
// else shares the device: a chunk waits only for chunks with smaller tickets, every one of which is held by a running
// workgroup that reaches it after chunks with still smaller tickets.  Same arithmetic order inside a chunk and the same
// chain words as scan_chunks_kernel<CHAINED>: identical bits.
#ifndef GLU_STREAM_GROUPS // (tuning builds override: tools/scan_chain_sweep.sh)
#define GLU_STREAM_GROUPS 4
#endif
constexpr int kStreamGroups = GLU_STREAM_GROUPS; // 16-byte groups per thread and buffer: 1024 x 4 x 4 = 16384 elements per chunk

// wave 0 of a workgroup: publishes the chunk's total, looks back for the carry-in, publishes the inclusive prefix; returns
// the carry-in (exclusive prefix of the chunk inside its partition) in every lane
template<typename T>
__device__ __forceinline__ T chain_resolve(unsigned long long* __restrict__ words, uint32_t chunk, uint32_t epoch, T total, uint32_t lane)
{
    union { T t; uint32_t u; } cv;
    cv.t = total;
    T prefix;
    {
        union { T t; uint32_t u; } z;
        z.u = 0;
        prefix = z.t;
    }
    if (chunk == 0)
    {
        if (lane == 0) __hip_atomic_store(&words[0], chain_pack(epoch, kChainGlobal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return prefix;
    }
    if (lane == 0) __hip_atomic_store(&words[chunk], chain_pack(epoch, kChainLocal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = (int) chunk - 1; // lane l examines chunk (look - l); the window moves 64 chunks at a time
    uint32_t spins = 0;
    for (;;)
    {
        const int idx = look - (int) lane;
        uint64_t w = chain_pack(epoch, kChainGlobal, 0u); // before the partition's first chunk: prefix 0
        if (idx >= 0) w = __hip_atomic_load(&words[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t flag = (w >> 32) & 3u;
        const bool ready = (uint32_t) (w >> 34) == epoch && flag != 0;
        const uint64_t ready_mask = __ballot(ready);
        const uint64_t global_mask = __ballot(ready && flag == kChainGlobal);
        uint64_t need = ~0ull; // lanes whose values are summed this round
        bool done = false;
        if (global_mask != 0)
        {
            const int g = __builtin_ctzll(global_mask); // nearest predecessor with an inclusive prefix
            need = g == 63 ? ~0ull : ((1ull << (g + 1)) - 1);
            done = true;
        }
        if ((ready_mask & need) == need)
        {
            union { T t; uint32_t u; } v, zero;
            v.u = (uint32_t) w;
            zero.u = 0;
            T part_sum = ((need >> lane) & 1ull) ? v.t : zero.t;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part_sum = combine<OP_SUM>(part_sum, shfl_down_t(part_sum, off));
            part_sum = shfl_t(part_sum, 0);
            prefix = combine<OP_SUM>(part_sum, prefix);
            if (done) break;
            look -= kW;
            spins = 0;
        }
        else
        {
            if (++spins > kChainSpinLimit) __builtin_trap(); // fail loudly, never hang
            __builtin_amdgcn_s_sleep(2);
        }
    }
    cv.t = combine<OP_SUM>(prefix, total);
    if (lane == 0) __hip_atomic_store(&words[chunk], chain_pack(epoch, kChainGlobal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prefix;
}

template<typename S, int N, bool ALIGNED, int GROUPS>
__device__ __forceinline__ void scan_chained_stream_body(Elem<S, N>* __restrict__ data, uint64_t count, uint32_t chunks,
                                                         uint32_t total_chunks, unsigned long long* __restrict__ chain,
                                                         uint32_t* __restrict__ ticket, uint32_t epoch)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T, GROUPS, kChainThreads>;
    static_assert(sizeof(T) == 4, "chained scan packs the value into 32 bits");
    __shared__ T wsum[2][C::WAVES];
    __shared__ uint32_t s_ticket[2];
    __shared__ T s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0)
    {
        s_ticket[0] = atomicAdd(ticket, 1u);
        s_ticket[1] = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    uint32_t t_cur = s_ticket[0], t_next = s_ticket[1];
    if (t_cur >= total_chunks) return; // (workgroup-uniform)

    auto chunk_base = [&](uint32_t t, uint32_t& valid) -> T* {
        const uint32_t chunk = t % chunks, part = t / chunks;
        const uint64_t cbeg = (uint64_t) chunk * C::CHUNK;
        valid = (count - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (count - cbeg) : (uint32_t) C::CHUNK;
        return data + (uint64_t) part * count + cbeg;
    };

    T xa[GROUPS][C::VEC], xb[GROUPS][C::VEC];
    {
        uint32_t valid;
        T* base = chunk_base(t_cur, valid);
        scan_load<S, N, ALIGNED, GROUPS>(base, valid, wave, lane, xa);
    }

    // one chunk: `cur` holds its elements (loads issued a step ago), the loads of the chunk after it go out first
    auto step = [&](T (&cur)[GROUPS][C::VEC], T (&nxt)[GROUPS][C::VEC], const uint32_t parity) {
        uint32_t requested = 0;
        if (tid == 0) requested = atomicAdd(ticket, 1u); // the ticket two chunks ahead; collected behind the look-back
        if (t_next < total_chunks)
        {
            uint32_t nvalid;
            T* nbase = chunk_base(t_next, nvalid);
            scan_load<S, N, ALIGNED, GROUPS>(nbase, nvalid, wave, lane, nxt);
        }
        uint32_t valid;
        T* base = chunk_base(t_cur, valid);
        const uint32_t chunk = t_cur % chunks, part = t_cur / chunks;

        T gexcl[GROUPS], gtot[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T lsum = cur[g][0];
#pragma unroll
            for (int k = 1; k < C::VEC; k++) lsum = combine<OP_SUM>(lsum, cur[g][k]);
            T incl = lsum;
#pragma unroll
            for (int off = 1; off < kW; off <<= 1)
            {
                T t = shfl_up_t(incl, off);
                if (lane >= (uint32_t) off) incl = combine<OP_SUM>(t, incl);
            }
            gtot[g] = shfl_t(incl, kW - 1);
            T up = shfl_up_t(incl, 1);
            gexcl[g] = lane == 0 ? zero_elem<S, N>() : up;
        }
        T wave_total = gtot[0];
#pragma unroll
        for (int g = 1; g < GROUPS; g++) wave_total = combine<OP_SUM>(wave_total, gtot[g]);
        if (lane == 0) wsum[parity][wave] = wave_total;
        __syncthreads();
        if (wave == 0)
        {
            T total = wsum[parity][0];
#pragma unroll
            for (int w = 1; w < C::WAVES; w++) total = combine<OP_SUM>(total, wsum[parity][w]);
            const T prefix = chain_resolve<T>(chain + (uint64_t) part * chunks, chunk, epoch, total, lane);
            if (lane == 0)
            {
                s_prefix = prefix;
                s_ticket[parity] = requested;
            }
        }
        __syncthreads();
        T run = s_prefix;
        const uint32_t t_after = s_ticket[parity];
#pragma unroll
        for (int w = 0; w < C::WAVES; w++)
            if ((uint32_t) w < wave) run = combine<OP_SUM>(run, wsum[parity][w]);
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T acc = combine<OP_SUM>(run, gexcl[g]);
            Pack<T, C::VEC> p;
#pragma unroll
            for (int k = 0; k < C::VEC; k++)
            {
                p.v[k] = acc;
                acc = combine<OP_SUM>(acc, cur[g][k]);
            }
            const uint32_t e0 = wave * C::WAVE_ELEMS + (g * kW + lane) * C::VEC;
            if (ALIGNED && e0 + C::VEC <= valid)
                *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p;
            else
            {
#pragma unroll
                for (int k = 0; k < C::VEC; k++)
                    if (e0 + k < valid) base[e0 + k] = p.v[k];
            }
            run = combine<OP_SUM>(run, gtot[g]);
        }
        t_cur = t_next;
        t_next = t_after;
    };
    for (;;)
    {
        step(xa, xb, 0u);
        if (t_cur >= total_chunks) break;
        step(xb, xa, 1u);
        if (t_cur >= total_chunks) break;
    }
}


} // namespace glu_hip

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

// Experimental copy of the chained kernel (uint32 only).  MODE 0: as the library's (ticket, local scan, publish, look-back).
// MODE 1: no look-back at all (carry-in 0: WRONG result -- the upper bound of anything a cheaper look-back could buy).
// MODE 2: the chunk's total is summed and published BEFORE the in-wave scans, which then run while the predecessors' words
// travel.  MODE 3: no ticket: chunks in blockIdx order (relies on in-order dispatch -- not something the library would ship
// without a proof; here to price the ticket).  MODE 4 = 2 + 3.
template<int GROUPS, int THREADS, int MODE, int MINW = 1>
__global__ __launch_bounds__(THREADS, MINW) void exp_chained_kernel(U* __restrict__ data, uint64_t count, uint32_t chunks,
                                                                     unsigned long long* __restrict__ chain, uint32_t* __restrict__ ticket,
                                                                     uint32_t epoch)
{
    using T = U;
    using C = ScanCfg<T, GROUPS, THREADS>;
    __shared__ T wsum[C::WAVES];
    __shared__ uint32_t s_ticket;
    __shared__ T s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t chunk = blockIdx.x;
    if (MODE != 3 && MODE != 4)
    {
        if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        chunk = s_ticket;
    }
    const uint64_t cbeg = (uint64_t) chunk * C::CHUNK;
    const uint32_t valid = (count - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (count - cbeg) : (uint32_t) C::CHUNK;
    T* base = data + cbeg;
    T x[GROUPS][C::VEC];
    scan_load<uint32_t, 1, true, GROUPS>(base, valid, wave, lane, x);
    T gexcl[GROUPS], gtot[GROUPS];
    auto local_scans = [&]() {
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T lsum = x[g][0];
#pragma unroll
            for (int k = 1; k < C::VEC; k++) lsum = combine<OP_SUM>(lsum, x[g][k]);
            T incl = lsum;
#pragma unroll
            for (int off = 1; off < kW; off <<= 1)
            {
                T t = shfl_up_t(incl, off);
                if (lane >= (uint32_t) off) incl = combine<OP_SUM>(t, incl);
            }
            gtot[g] = shfl_t(incl, kW - 1);
            T up = shfl_up_t(incl, 1);
            gexcl[g] = lane == 0 ? zero_elem<uint32_t, 1>() : up;
        }
    };
    if (MODE == 2 || MODE == 4)
    {
        T acc = zero_elem<uint32_t, 1>();
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
#pragma unroll
            for (int k = 0; k < C::VEC; k++) acc = combine<OP_SUM>(acc, x[g][k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc = combine<OP_SUM>(acc, shfl_down_t(acc, off));
        if (lane == 0) wsum[wave] = acc;
    }
    else
    {
        local_scans();
        T wave_total = gtot[0];
#pragma unroll
        for (int g = 1; g < GROUPS; g++) wave_total = combine<OP_SUM>(wave_total, gtot[g]);
        if (lane == 0) wsum[wave] = wave_total;
    }
    __syncthreads();
    if (wave == 0)
    {
        T total = wsum[0];
#pragma unroll
        for (int w = 1; w < C::WAVES; w++) total = combine<OP_SUM>(total, wsum[w]);
        T prefix = zero_elem<uint32_t, 1>();
        if (MODE == 1)
        {
            if (lane == 0) __hip_atomic_store(&chain[chunk], chain_pack(epoch, kChainGlobal, total.c[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
            prefix = chain_resolve<T>(chain, chunk, epoch, total, lane);
        if (lane == 0) s_prefix = prefix;
    }
    if (MODE == 2 || MODE == 4) local_scans(); // (wave 0 after its look-back, the others while it looks back)
    __syncthreads();
    T run = s_prefix;
#pragma unroll
    for (int w = 0; w < C::WAVES; w++)
        if ((uint32_t) w < wave) run = combine<OP_SUM>(run, wsum[w]);
#pragma unroll
    for (int g = 0; g < GROUPS; g++)
    {
        T acc = combine<OP_SUM>(run, gexcl[g]);
        Pack<T, C::VEC> p;
#pragma unroll
        for (int k = 0; k < C::VEC; k++)
        {
            p.v[k] = acc;
            acc = combine<OP_SUM>(acc, x[g][k]);
        }
        const uint32_t e0 = wave * C::WAVE_ELEMS + (g * kW + lane) * C::VEC;
        if (e0 + C::VEC <= valid) *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p;
        run = combine<OP_SUM>(run, gtot[g]);
    }
}

// Persistent, SINGLE-buffered: a workgroup stays and takes chunk after chunk in ticket order; the ticket of the next chunk is
// requested at the start of a chunk and collected behind its look-back, and the next chunk's loads go out right behind the
// current chunk's stores, into the same registers (a store has read its registers when it is issued).  Against one workgroup
// per chunk this saves the workgroup's exit, the dispatch of the next one and the exposed ticket round trip, and keeps whole
// 8-group chunks in flight (two workgroups per CU as before).
template<int GROUPS, int THREADS, int MINW = 1>
__global__ __launch_bounds__(THREADS, MINW) void persist_chained_kernel(U* __restrict__ data, uint64_t count, uint32_t chunks,
                                                                         unsigned long long* __restrict__ chain, uint32_t* __restrict__ ticket,
                                                                         uint32_t epoch)
{
    using T = U;
    using C = ScanCfg<T, GROUPS, THREADS>;
    __shared__ T wsum[2][C::WAVES];
    __shared__ uint32_t s_ticket[2];
    __shared__ T s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket[1] = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t chunk = s_ticket[1];
    uint32_t parity = 0;
    while (chunk < chunks)
    {
        uint32_t requested = 0;
        if (tid == 0) requested = atomicAdd(ticket, 1u);
        const uint64_t cbeg = (uint64_t) chunk * C::CHUNK;
        const uint32_t valid = (count - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (count - cbeg) : (uint32_t) C::CHUNK;
        T* base = data + cbeg;
        T x[GROUPS][C::VEC];
        scan_load<uint32_t, 1, true, GROUPS>(base, valid, wave, lane, x);
        T gexcl[GROUPS], gtot[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T lsum = x[g][0];
#pragma unroll
            for (int k = 1; k < C::VEC; k++) lsum = combine<OP_SUM>(lsum, x[g][k]);
            T incl = lsum;
#pragma unroll
            for (int off = 1; off < kW; off <<= 1)
            {
                T t = shfl_up_t(incl, off);
                if (lane >= (uint32_t) off) incl = combine<OP_SUM>(t, incl);
            }
            gtot[g] = shfl_t(incl, kW - 1);
            T up = shfl_up_t(incl, 1);
            gexcl[g] = lane == 0 ? zero_elem<uint32_t, 1>() : up;
        }
        T wave_total = gtot[0];
#pragma unroll
        for (int g = 1; g < GROUPS; g++) wave_total = combine<OP_SUM>(wave_total, gtot[g]);
        if (lane == 0) wsum[parity][wave] = wave_total;
        __syncthreads();
        if (wave == 0)
        {
            T total = wsum[parity][0];
#pragma unroll
            for (int w = 1; w < C::WAVES; w++) total = combine<OP_SUM>(total, wsum[parity][w]);
            const T prefix = chain_resolve<T>(chain, chunk, epoch, total, lane);
            if (lane == 0)
            {
                s_prefix = prefix;
                s_ticket[parity] = requested;
            }
        }
        __syncthreads();
        T run = s_prefix;
        const uint32_t next = s_ticket[parity];
#pragma unroll
        for (int w = 0; w < C::WAVES; w++)
            if ((uint32_t) w < wave) run = combine<OP_SUM>(run, wsum[parity][w]);
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T acc = combine<OP_SUM>(run, gexcl[g]);
            Pack<T, C::VEC> p;
#pragma unroll
            for (int k = 0; k < C::VEC; k++)
            {
                p.v[k] = acc;
                acc = combine<OP_SUM>(acc, x[g][k]);
            }
            const uint32_t e0 = wave * C::WAVE_ELEMS + (g * kW + lane) * C::VEC;
            if (e0 + C::VEC <= valid) *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p;
            run = combine<OP_SUM>(run, gtot[g]);
        }
        chunk = next;
        parity ^= 1u;
    }
}

// LDS-DMA prefetch (round 4, the last idea of DESIGN.md section 8): persistent workgroups, one per CU; the NEXT chunk streams
// into LDS with global_load_lds_dwordx4 (no registers) -- issued by waves 1 .. 15 at the start of a step, so that wave 0's
// look-back loads never queue behind them -- while the current chunk is scanned in registers, its carry-in is looked up and its
// stores go out; then the next chunk moves LDS -> registers (ds_read_b128).  Loads and stores of a CU overlap all the time, the
// look-back is off the memory pipe's critical path.  Raw s_barrier + lgkmcnt(0) (a __syncthreads() would drain the DMA).
// count must be a multiple of the chunk (tuning harness).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// STATIC: no tickets -- workgroup b takes the chunks b, b + grid, b + 2 grid, ... (all workgroups must be resident: an experiment)
template<int GROUPS, int PREFETCH_WAVES = 15, bool STATIC = false>
__global__ __launch_bounds__(1024) void lds_chained_kernel(U* __restrict__ data, uint64_t count, uint32_t chunks,
                                                           unsigned long long* __restrict__ chain, uint32_t* __restrict__ ticket,
                                                           uint32_t epoch)
{
    using T = U;
    using C = ScanCfg<T, GROUPS, 1024>;
    constexpr int PIECES = GROUPS * 16; // 1 KiB each: one wave-instruction of global_load_lds_dwordx4
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[]; // the next chunk: PIECES KiB
    __shared__ T wsum[2][C::WAVES];
    __shared__ uint32_t s_ticket[2];
    __shared__ T s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    auto prefetch = [&](uint32_t t) {
        if (wave == 0 || wave > (uint32_t) PREFETCH_WAVES) return; // wave 0 keeps its memory queue for the look-back
        const unsigned char* gbase = reinterpret_cast<const unsigned char*>(data + (uint64_t) t * C::CHUNK) + lane * 16;
        for (int p = (int) wave - 1; p < PIECES; p += PREFETCH_WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (gbase + (size_t) p * 1024),
                                             (__attribute__((address_space(3))) void*) (lds + p * 1024), 16, 0, 0);
    };
    T x[GROUPS][C::VEC];
    auto read_chunk = [&]() {
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + (wave * GROUPS + g) * 1024 + lane * 16);
            x[g][0].c[0] = v.x, x[g][1].c[0] = v.y, x[g][2].c[0] = v.z, x[g][3].c[0] = v.w;
        }
    };

    if (!STATIC && tid == 0)
    {
        s_ticket[0] = atomicAdd(ticket, 1u);
        s_ticket[1] = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    uint32_t t_cur = STATIC ? blockIdx.x : s_ticket[0], t_next = STATIC ? blockIdx.x + gridDim.x : s_ticket[1];
    if (t_cur >= chunks) return;
    prefetch(t_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    read_chunk();
    lds_barrier(); // the LDS buffer is free
    uint32_t parity = 0;
    for (;;)
    {
        if (t_next < chunks) prefetch(t_next);
        uint32_t requested = 0;
        if (!STATIC && tid == 0) requested = atomicAdd(ticket, 1u);
        T* base = data + (uint64_t) t_cur * C::CHUNK;
        T gexcl[GROUPS], gtot[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T lsum = x[g][0];
#pragma unroll
            for (int k = 1; k < C::VEC; k++) lsum = combine<OP_SUM>(lsum, x[g][k]);
            T incl = lsum;
#pragma unroll
            for (int off = 1; off < kW; off <<= 1)
            {
                T t = shfl_up_t(incl, off);
                if (lane >= (uint32_t) off) incl = combine<OP_SUM>(t, incl);
            }
            gtot[g] = shfl_t(incl, kW - 1);
            T up = shfl_up_t(incl, 1);
            gexcl[g] = lane == 0 ? zero_elem<uint32_t, 1>() : up;
        }
        T wave_total = gtot[0];
#pragma unroll
        for (int g = 1; g < GROUPS; g++) wave_total = combine<OP_SUM>(wave_total, gtot[g]);
        if (lane == 0) wsum[parity][wave] = wave_total;
        lds_barrier();
        if (wave == 0)
        {
            T total = wsum[parity][0];
#pragma unroll
            for (int w = 1; w < C::WAVES; w++) total = combine<OP_SUM>(total, wsum[parity][w]);
            const T prefix = chain_resolve<T>(chain, t_cur, epoch, total, lane);
            if (lane == 0)
            {
                s_prefix = prefix;
                s_ticket[parity] = requested;
            }
        }
        lds_barrier();
        T run = s_prefix;
        const uint32_t t_after = STATIC ? t_next + gridDim.x : s_ticket[parity];
#pragma unroll
        for (int w = 0; w < C::WAVES; w++)
            if ((uint32_t) w < wave) run = combine<OP_SUM>(run, wsum[parity][w]);
        // the next chunk has landed in LDS (issued a scan and a look-back ago); the stores of the chunk before have long been
        // acknowledged: nothing to wait for in the common case
        if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < GROUPS; g++)
        {
            T acc = combine<OP_SUM>(run, gexcl[g]);
            Pack<T, C::VEC> p;
#pragma unroll
            for (int k = 0; k < C::VEC; k++)
            {
                p.v[k] = acc;
                acc = combine<OP_SUM>(acc, x[g][k]);
            }
            const uint32_t e0 = wave * C::WAVE_ELEMS + (g * kW + lane) * C::VEC;
            *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p;
            run = combine<OP_SUM>(run, gtot[g]);
        }
        t_cur = t_next;
        t_next = t_after;
        parity ^= 1u;
        if (t_cur >= chunks) break;
        lds_barrier(); // every wave's share of the prefetch has landed
        read_chunk();
        lds_barrier(); // the LDS buffer is free for the next prefetch
    }
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
    CK(hipMalloc(&chain, (n / 2048 + 16) * 8));
    CK(hipMalloc(&ticket, 256));
    CK(hipMemset(chain, 0, (n / 2048 + 16) * 8));
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
    if (getenv("SSB_STREAM"))
    {
        STREAM(4, 4, 1)
        STREAM(4, 4, 2)
        STREAM(8, 4, 1)
    }
#define EXP(G, TH, MODE, NOTE)                                                                                                    \
    {                                                                                                                            \
        using CE = ScanCfg<U, G, TH>;                                                                                            \
        const uint32_t ch = (uint32_t) ((n + CE::CHUNK - 1) / CE::CHUNK);                                                        \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "%s%d threads x %d groups, mode %d %s", MODE == 1 ? "~ " : "", TH, G, MODE, NOTE);            \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL((exp_chained_kernel<G, TH, MODE>), dim3(ch), dim3(TH), 0, 0, (U*) work, (uint64_t) n, ch, chain, ticket, epoch); \
        });                                                                                                                      \
    }
#define PERSIST(G, TH, MINW, WGS)                                                                                                 \
    {                                                                                                                            \
        using CE = ScanCfg<U, G, TH>;                                                                                            \
        const uint32_t ch = (uint32_t) ((n + CE::CHUNK - 1) / CE::CHUNK);                                                        \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "persistent single-buffered: %d x %d groups, min waves/EU %d, %d workgroups/CU", TH, G, MINW, WGS); \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL((persist_chained_kernel<G, TH, MINW>), dim3(std::min<uint32_t>(ch, cus* WGS)), dim3(TH), 0, 0, (U*) work, \
                               (uint64_t) n, ch, chain, ticket, epoch);                                                          \
        });                                                                                                                      \
    }
#define LDSV2(G, PW, WGS)                                                                                                         \
    {                                                                                                                            \
        using CE = ScanCfg<U, G, 1024>;                                                                                          \
        const uint32_t ch = (uint32_t) (n / CE::CHUNK);                                                                          \
        auto kern = lds_chained_kernel<G, PW>;                                                                                   \
        CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, G * 16 * 1024));                  \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "LDS-DMA prefetch: 1024 x %d groups, %d prefetching waves, %d workgroups/CU", G, PW, WGS);    \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL(kern, dim3(std::min<uint32_t>(ch, cus * WGS)), dim3(1024), G * 16 * 1024, 0, (U*) work, (uint64_t) n, ch, chain, ticket, epoch); \
        });                                                                                                                      \
    }
#define LDSV(G, PW)                                                                                                               \
    {                                                                                                                            \
        using CE = ScanCfg<U, G, 1024>;                                                                                          \
        const uint32_t ch = (uint32_t) (n / CE::CHUNK);                                                                          \
        auto kern = lds_chained_kernel<G, PW>;                                                                                   \
        CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, G * 16 * 1024));                  \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "LDS-DMA prefetch: 1024 x %d groups, %d prefetching waves, 1 workgroup/CU", G, PW);           \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL(kern, dim3(std::min<uint32_t>(ch, cus)), dim3(1024), G * 16 * 1024, 0, (U*) work, (uint64_t) n, ch, chain, ticket, epoch); \
        });                                                                                                                      \
    }
    LDSV(8, 15)
    LDSV(8, 8)
    LDSV(9, 15)
    LDSV(4, 15)
    LDSV(6, 15)
    LDSV2(4, 15, 2)
    {
        using CE = ScanCfg<U, 8, 1024>;
        const uint32_t ch = (uint32_t) (n / CE::CHUNK);
        auto kern = lds_chained_kernel<8, 15, true>;
        CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 1024));
        run("LDS-DMA prefetch, STATIC chunk assignment (no tickets), 1024 x 8, 1 workgroup/CU", [&] {
            hipLaunchKernelGGL(kern, dim3(std::min<uint32_t>(ch, cus)), dim3(1024), 8 * 16 * 1024, 0, (U*) work, (uint64_t) n, ch, chain, ticket, epoch);
        });
    }
    if (getenv("SSB_ONLY_LDS")) return 0;
    PERSIST(8, 1024, 8, 2)
    PERSIST(8, 1024, 4, 2)
    PERSIST(8, 1024, 4, 1)
    PERSIST(16, 1024, 4, 1)
    PERSIST(4, 1024, 8, 2)
    PERSIST(8, 512, 8, 4)
    PERSIST(16, 512, 4, 2)
#define EXPW(G, TH, MODE, MINW, NOTE)                                                                                             \
    {                                                                                                                            \
        using CE = ScanCfg<U, G, TH>;                                                                                            \
        const uint32_t ch = (uint32_t) ((n + CE::CHUNK - 1) / CE::CHUNK);                                                        \
        char name[96];                                                                                                           \
        snprintf(name, sizeof name, "%d threads x %d groups, mode %d, min waves/EU %d %s", TH, G, MODE, MINW, NOTE);              \
        run(name, [&] {                                                                                                          \
            hipLaunchKernelGGL((exp_chained_kernel<G, TH, MODE, MINW>), dim3(ch), dim3(TH), 0, 0, (U*) work, (uint64_t) n, ch, chain, ticket, epoch); \
        });                                                                                                                      \
    }
    EXPW(8, 1024, 0, 8, "(64 VGPRs: two workgroups per CU)")
    EXPW(8, 1024, 0, 4, "")
    EXPW(6, 1024, 0, 8, "")
    EXPW(7, 1024, 0, 8, "")
    EXPW(10, 1024, 0, 4, "")
    EXPW(12, 1024, 0, 4, "")
    EXPW(8, 512, 0, 8, "(64 VGPRs: four workgroups per CU)")
    EXPW(12, 512, 0, 6, "")
    EXPW(16, 512, 0, 4, "")
    if (getenv("SSB_ONLY_PERSIST")) return 0;
    EXP(8, 1024, 0, "(replica of the library kernel)")
    EXP(8, 1024, 1, "(NO look-back: wrong result, upper bound)")
    EXP(8, 1024, 2, "(total published before the scans)")
    EXP(8, 1024, 3, "(blockIdx order, no ticket)")
    EXP(8, 1024, 4, "(early total + blockIdx order)")
    EXP(8, 512, 0, "")
    EXP(8, 512, 2, "")
    EXP(8, 512, 4, "")
    EXP(4, 512, 4, "")
    EXP(8, 256, 4, "")
    EXP(4, 256, 4, "")
    EXP(4, 256, 1, "(NO look-back, tickets)")
    EXP(4, 256, 0, "")
    EXP(16, 256, 4, "")
    EXP(16, 512, 4, "")
    return 0;
}
