// scan_reduce_kernels.hpp -- gfx950 kernels behind glu::BlellochScan and glu::Reduce.
//
//   exclusive scan  <- upsweep/downsweep shaders (reference glu/BlellochScan.hpp:13-76): same result
//                      (in-place exclusive `+` scan of adjacent partitions), computed as
//                      chunk sums -> scan of the chunk sums (recursive) -> chunk scan with carry-in,
//                      i.e. 3 launches for 2^28 elements instead of 2*log2(count) dispatches.
//   reduce          <- reduction shader (reference glu/Reduce.hpp:11-38): data[0] = op over data[0..count),
//                      wave64 shuffles + LDS, two launches; no subgroup-size-32 assumption.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_utils.hpp"

namespace glu_hip
{
constexpr int kW = 64;

// ---- element types: scalar S or N-component vector of S, std430 strides (4/8/16/32 B) ---------------------
template<typename S, int N>
struct alignas((sizeof(S) * N >= 16) ? 16 : sizeof(S) * N) Elem
{
    S c[N];
};

enum OpKind
{
    OP_SUM = 0,
    OP_MUL = 1,
    OP_MIN = 2,
    OP_MAX = 3
};

template<typename S>
struct ScalarOps
{
    using U = S;
    static __device__ __forceinline__ S sum(S a, S b) { return a + b; }
    static __device__ __forceinline__ S mul(S a, S b) { return a * b; }
    static __device__ __forceinline__ S mn(S a, S b) { return a < b ? a : b; }
    static __device__ __forceinline__ S mx(S a, S b) { return a > b ? a : b; }
};
template<>
struct ScalarOps<int32_t>
{ // GLSL int arithmetic wraps; do it unsigned to keep C++ well-defined
    static __device__ __forceinline__ int32_t sum(int32_t a, int32_t b) { return (int32_t) ((uint32_t) a + (uint32_t) b); }
    static __device__ __forceinline__ int32_t mul(int32_t a, int32_t b) { return (int32_t) ((uint32_t) a * (uint32_t) b); }
    static __device__ __forceinline__ int32_t mn(int32_t a, int32_t b) { return a < b ? a : b; }
    static __device__ __forceinline__ int32_t mx(int32_t a, int32_t b) { return a > b ? a : b; }
};

template<int OP, typename S, int N>
__device__ __forceinline__ Elem<S, N> combine(const Elem<S, N>& a, const Elem<S, N>& b)
{
    Elem<S, N> r;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        if (OP == OP_SUM) r.c[i] = ScalarOps<S>::sum(a.c[i], b.c[i]);
        else if (OP == OP_MUL) r.c[i] = ScalarOps<S>::mul(a.c[i], b.c[i]);
        else if (OP == OP_MIN) r.c[i] = ScalarOps<S>::mn(a.c[i], b.c[i]);
        else r.c[i] = ScalarOps<S>::mx(a.c[i], b.c[i]);
    }
    return r;
}

template<typename S, int N>
__device__ __forceinline__ Elem<S, N> zero_elem()
{
    Elem<S, N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.c[i] = (S) 0;
    return r;
}

// cross-lane moves of an arbitrary Elem, one dword at a time
template<typename T>
__device__ __forceinline__ T shfl_up_t(const T& v, int delta)
{
    constexpr int W = sizeof(T) / 4;
    union { T t; uint32_t w[W]; } in, out;
    in.t = v;
#pragma unroll
    for (int i = 0; i < W; i++) out.w[i] = __shfl_up(in.w[i], delta, kW);
    return out.t;
}
template<typename T>
__device__ __forceinline__ T shfl_down_t(const T& v, int delta)
{
    constexpr int W = sizeof(T) / 4;
    union { T t; uint32_t w[W]; } in, out;
    in.t = v;
#pragma unroll
    for (int i = 0; i < W; i++) out.w[i] = __shfl_down(in.w[i], delta, kW);
    return out.t;
}
template<typename T>
__device__ __forceinline__ T shfl_t(const T& v, int src)
{
    constexpr int W = sizeof(T) / 4;
    union { T t; uint32_t w[W]; } in, out;
    in.t = v;
#pragma unroll
    for (int i = 0; i < W; i++) out.w[i] = __shfl(in.w[i], src, kW);
    return out.t;
}

// ---- scan -------------------------------------------------------------------------------------------------
// A chunk = THREADS * GROUPS * VEC elements.  Inside a wave the layout is "group-major, then lane, then the VEC
// elements of a 16-byte vector", so every load/store instruction of a wave is one contiguous 1 KiB.
// GROUPS_ = 4, THREADS_ = 256: 4096 4-byte elements per chunk (reduce-then-scan path); the chained path uses 1024
// threads x 8 groups (32768 elements, 128 KiB per workgroup) so that the single ticket counter sees 8 K atomics for 2^28
// elements instead of 65 K (one global counter sustains only ~90 returning atomics per microsecond on MI355X).
template<typename T, int GROUPS_ = 4, int THREADS_ = 256>
struct ScanCfg
{
    static constexpr int THREADS = THREADS_;
    static constexpr int WAVES = THREADS / kW;
    static constexpr int VEC = sizeof(T) >= 16 ? 1 : 16 / (int) sizeof(T);
    static constexpr int GROUPS = GROUPS_;
    static constexpr int WAVE_ELEMS = kW * GROUPS * VEC;
    static constexpr int CHUNK = WAVES * WAVE_ELEMS;
};

template<typename T, int VEC>
struct alignas(sizeof(T) * VEC >= 16 ? 16 : sizeof(T) * VEC) Pack
{
    T v[VEC];
};

// Loads the calling wave's elements of one chunk.  `base` points at the chunk start, `valid` = elements of the
// chunk that exist (the rest read as zero).
template<typename S, int N, bool ALIGNED, int GROUPS = 4>
__device__ __forceinline__ void scan_load(const Elem<S, N>* base, uint32_t valid, uint32_t wave, uint32_t lane,
                                          Elem<S, N> (&x)[GROUPS][ScanCfg<Elem<S, N>>::VEC])
{
    using T = Elem<S, N>;
    using C = ScanCfg<T, GROUPS>; // only WAVE_ELEMS / VEC are used here: independent of the thread count
#pragma unroll
    for (int g = 0; g < C::GROUPS; g++)
    {
        const uint32_t e0 = wave * C::WAVE_ELEMS + (g * kW + lane) * C::VEC;
        if (ALIGNED && e0 + C::VEC <= valid)
        {
            Pack<T, C::VEC> p = *reinterpret_cast<const Pack<T, C::VEC>*>(base + e0); // (non-temporal: slower, the chunk is rewritten in place)
#pragma unroll
            for (int k = 0; k < C::VEC; k++) x[g][k] = p.v[k];
        }
        else
        {
#pragma unroll
            for (int k = 0; k < C::VEC; k++) x[g][k] = (e0 + k < valid) ? base[e0 + k] : zero_elem<S, N>();
        }
    }
}

// sums[partition * chunks + chunk] = sum of the chunk
template<typename S, int N, bool ALIGNED>
__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const Elem<S, N>* __restrict__ data,
                                                              Elem<S, N>* __restrict__ sums, uint64_t count,
                                                              uint32_t chunks)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T>;
    __shared__ T wsum[C::WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t chunk = blockIdx.x % chunks, part = blockIdx.x / chunks; // grid = partitions * chunks
    const uint64_t cbeg = (uint64_t) chunk * C::CHUNK;
    const uint32_t valid = (count - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (count - cbeg) : (uint32_t) C::CHUNK;
    const T* base = data + (uint64_t) part * count + cbeg;

    T x[C::GROUPS][C::VEC];
    scan_load<S, N, ALIGNED>(base, valid, wave, lane, x);
    T acc = zero_elem<S, N>();
#pragma unroll
    for (int g = 0; g < C::GROUPS; g++)
#pragma unroll
        for (int k = 0; k < C::VEC; k++) acc = combine<OP_SUM>(acc, x[g][k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = combine<OP_SUM>(acc, shfl_down_t(acc, off));
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (tid == 0)
    {
        T t = wsum[0];
#pragma unroll
        for (int w = 1; w < C::WAVES; w++) t = combine<OP_SUM>(t, wsum[w]);
        sums[(uint64_t) part * chunks + chunk] = t;
    }
}

// ---- chained scan state (single-pass "decoupled look-back") ---------------------------------------------------
// One 64-bit word per chunk, written and read with agent-scope relaxed atomics (one aligned 8-byte store: the data IS
// the flag, no separate release/acquire needed): [63:34] epoch of the launch, [33:32] 1 = chunk total, 2 = inclusive
// prefix up to and including the chunk, [31:0] the 4-byte value.  Words of older epochs read as "not ready", so the
// array is zeroed once at allocation, never per launch.
constexpr uint64_t kChainLocal = 1, kChainGlobal = 2;
__device__ __forceinline__ uint64_t chain_pack(uint32_t epoch, uint64_t flag, uint32_t value)
{
    return ((uint64_t) epoch << 34) | (flag << 32) | value;
}
#ifndef GLU_CHAIN_THREADS // (tuning builds override the two: tools/scan_chain_sweep.sh)
#define GLU_CHAIN_THREADS 1024
#endif
#ifndef GLU_CHAIN_GROUPS
#define GLU_CHAIN_GROUPS 8
#endif
constexpr int kChainThreads = GLU_CHAIN_THREADS;
constexpr int kChainMinChunks = 256;
constexpr int kChainGroups = GLU_CHAIN_GROUPS; // 16-byte load groups per thread in the chained kernel
constexpr uint32_t kChainSpinLimit = 1u << 24; // polls before the kernel gives up loudly (trap) instead of hanging

// In-place exclusive scan of every chunk.
//   CHAINED = false: carry-in from carry[partition * chunks + chunk] (nullptr = 0): the reduce-then-scan path.
//   CHAINED = true (4-byte element types): single pass.  Chunks are taken in ticket order (so every chunk a
//   workgroup waits for is already running), each publishes its total, looks back over its predecessors' words for
//   the carry-in, then publishes its inclusive prefix.  8 B/element of HBM traffic instead of 12.
template<typename S, int N, bool ALIGNED, bool CHAINED = false>
__global__ __launch_bounds__(CHAINED ? kChainThreads : 256) void scan_chunks_kernel(Elem<S, N>* __restrict__ data,
                                                          const Elem<S, N>* __restrict__ carry, uint64_t count,
                                                          uint32_t chunks, unsigned long long* __restrict__ chain = nullptr,
                                                          uint32_t* __restrict__ ticket = nullptr, uint32_t epoch = 0)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T, CHAINED ? kChainGroups : 4, CHAINED ? kChainThreads : 256>;
    static_assert(!CHAINED || sizeof(T) == 4, "chained scan packs the value into 32 bits");
    __shared__ T wsum[C::WAVES];
    __shared__ uint32_t s_ticket;
    __shared__ T s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t linear = blockIdx.x;
    if (CHAINED)
    {
        if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        linear = s_ticket;
    }
    const uint32_t chunk = linear % chunks, part = linear / chunks; // grid = partitions * chunks
    const uint64_t cbeg = (uint64_t) chunk * C::CHUNK;
    const uint32_t valid = (count - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (count - cbeg) : (uint32_t) C::CHUNK;
    T* base = data + (uint64_t) part * count + cbeg;

    T x[C::GROUPS][C::VEC];
    scan_load<S, N, ALIGNED, C::GROUPS>(base, valid, wave, lane, x);

    // per group: lane-local exclusive over the VEC elements, then a wave scan of the lane sums
    T gexcl[C::GROUPS]; // exclusive prefix of this lane inside its group
    T gtot[C::GROUPS];  // group totals (wave-uniform)
#pragma unroll
    for (int g = 0; g < C::GROUPS; g++)
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
        gexcl[g] = lane == 0 ? zero_elem<S, N>() : up;
    }
    T wave_total = gtot[0];
#pragma unroll
    for (int g = 1; g < C::GROUPS; g++) wave_total = combine<OP_SUM>(wave_total, gtot[g]);
    if (lane == 0) wsum[wave] = wave_total;
    __syncthreads();

    T run = zero_elem<S, N>();
    if (CHAINED)
    {
        if (wave == 0)
        {
            T total = wsum[0];
#pragma unroll
            for (int w = 1; w < C::WAVES; w++) total = combine<OP_SUM>(total, wsum[w]);
            union { T t; uint32_t u; } cv;
            unsigned long long* words = chain + (uint64_t) part * chunks;
            T prefix = zero_elem<S, N>();
            if (chunk == 0)
            {
                cv.t = total;
                if (lane == 0)
                    __hip_atomic_store(&words[0], chain_pack(epoch, kChainGlobal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
                cv.t = total;
                if (lane == 0)
                    __hip_atomic_store(&words[chunk], chain_pack(epoch, kChainLocal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // look back: lane l examines chunk (look - l); the window moves 64 chunks at a time
                int look = (int) chunk - 1;
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
                        union { T t; uint32_t u; } v;
                        v.u = (uint32_t) w;
                        T part_sum = ((need >> lane) & 1ull) ? v.t : zero_elem<S, N>();
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
                if (lane == 0)
                    __hip_atomic_store(&words[chunk], chain_pack(epoch, kChainGlobal, cv.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_prefix = prefix;
        }
        __syncthreads();
        run = s_prefix;
    }
    else if (carry)
    {
        run = carry[(uint64_t) part * chunks + chunk];
    }
#pragma unroll
    for (int w = 0; w < C::WAVES; w++)
        if ((uint32_t) w < wave) run = combine<OP_SUM>(run, wsum[w]);

#pragma unroll
    for (int g = 0; g < C::GROUPS; g++)
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
        if (ALIGNED && e0 + C::VEC <= valid)
        {
            *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p; // (non-temporal stores: no gain at 2^28, slower at 2^26)
        }
        else
        {
#pragma unroll
            for (int k = 0; k < C::VEC; k++)
                if (e0 + k < valid) base[e0 + k] = p.v[k];
        }
        run = combine<OP_SUM>(run, gtot[g]);
    }
}

// MANY SMALL PARTITIONS: count is a power of two and at most one wave's span of a chunk (ScanCfg::WAVE_ELEMS: 1024 elements
// of 4 bytes), so no partition crosses a wave.  A workgroup takes CHUNK consecutive elements of the whole array -- CHUNK / count
// partitions -- instead of one partition (a 256-element partition would use one of its 16 wave-groups: 2^18 partitions of 256
// elements took 387 us, the same 2^26 elements as one partition 104 us).  Same arithmetic order as scan_chunks_kernel inside a
// partition (lane-local, wave scan of the lane sums, running sum over the wave's groups), so the results are the same bits.
//   count >= 64 * VEC (a whole group or more): plain wave scan per group, the running sum restarts at partition starts;
//   VEC < count < 64 * VEC: segmented wave scan, count / VEC lanes per partition;
//   count <= VEC: inside one lane's vector.
template<typename S, int N, bool ALIGNED>
__global__ __launch_bounds__(256) void scan_small_partitions_kernel(Elem<S, N>* __restrict__ data, uint64_t total, uint32_t count)
{
    using T = Elem<S, N>;
    using C = ScanCfg<T>;
    constexpr uint32_t GS = kW * C::VEC; // elements of one group
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t cbeg = (uint64_t) blockIdx.x * C::CHUNK;
    const uint32_t valid = (total - cbeg) < (uint64_t) C::CHUNK ? (uint32_t) (total - cbeg) : (uint32_t) C::CHUNK;
    T* base = data + cbeg;

    T x[C::GROUPS][C::VEC];
    scan_load<S, N, ALIGNED, C::GROUPS>(base, valid, wave, lane, x);

    const uint32_t seg_lanes = count > (uint32_t) C::VEC ? count / C::VEC : 1u; // lanes of one partition (power of two)
    const uint32_t pos = seg_lanes >= (uint32_t) kW ? lane : (lane & (seg_lanes - 1u));
    T run = zero_elem<S, N>();
#pragma unroll
    for (int g = 0; g < C::GROUPS; g++)
    {
        // lane-local exclusive prefixes; partitions shorter than the vector restart inside it
        T loc[C::VEC];
        T acc = zero_elem<S, N>();
#pragma unroll
        for (int k = 0; k < C::VEC; k++)
        {
            if (count < (uint32_t) C::VEC && ((uint32_t) k & (count - 1u)) == 0) acc = zero_elem<S, N>();
            loc[k] = acc;
            acc = combine<OP_SUM>(acc, x[g][k]);
        }
        T excl = zero_elem<S, N>();
        T gtot = zero_elem<S, N>();
        if (count > (uint32_t) C::VEC) // (kernel-uniform)
        {
            T incl = acc;
#pragma unroll
            for (int off = 1; off < kW; off <<= 1)
            {
                T t = shfl_up_t(incl, off);
                if ((uint32_t) off < seg_lanes && pos >= (uint32_t) off) incl = combine<OP_SUM>(t, incl);
            }
            gtot = shfl_t(incl, kW - 1); // (used when a partition is at least a group)
            T up = shfl_up_t(incl, 1);
            if (pos != 0) excl = up;
        }
        const uint32_t gbeg = wave * C::WAVE_ELEMS + g * GS;
        if (count < GS || (gbeg & (count - 1u)) == 0) run = zero_elem<S, N>();
        T out = combine<OP_SUM>(run, excl);
        Pack<T, C::VEC> p;
#pragma unroll
        for (int k = 0; k < C::VEC; k++) p.v[k] = combine<OP_SUM>(out, loc[k]);
        const uint32_t e0 = gbeg + lane * C::VEC;
        if (ALIGNED && e0 + C::VEC <= valid)
            *reinterpret_cast<Pack<T, C::VEC>*>(base + e0) = p;
        else
        {
#pragma unroll
            for (int k = 0; k < C::VEC; k++)
                if (e0 + k < valid) base[e0 + k] = p.v[k];
        }
        run = combine<OP_SUM>(run, gtot);
    }
}

// ---- reduce -----------------------------------------------------------------------------------------------
template<int OP, typename S, int N>
__device__ __forceinline__ Elem<S, N> block_reduce(Elem<S, N> acc, bool has, Elem<S, N>* wtmp, uint32_t* whas,
                                                   uint32_t tid, bool& out_has)
{
    // lanes without any element are skipped (no identity element needed for min/max/mul)
    using T = Elem<S, N>;
    const uint32_t lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        T o = shfl_down_t(acc, off);
        int oh = __shfl_down((int) has, off, kW);
        bool other_ok = (lane + off < (uint32_t) kW) && oh;
        if (other_ok) acc = has ? combine<OP>(acc, o) : o;
        has = has || other_ok;
    }
    if (lane == 0)
    {
        wtmp[wave] = acc;
        whas[wave] = has;
    }
    __syncthreads();
    T r = wtmp[0];
    bool rh = whas[0];
    const uint32_t nw = blockDim.x / kW;
    for (uint32_t w = 1; w < nw; w++)
    {
        if (whas[w])
        {
            r = rh ? combine<OP>(r, wtmp[w]) : wtmp[w];
            rh = true;
        }
    }
    out_has = rh;
    return r;
}

// out[b] = op over workgroup b's grid-strided share of in[0..count).  Launch with gridDim.x <= max(1, packs / 256)
// so that every workgroup owns at least one element; with gridDim.x == 1 the result lands in out[0] (out may
// alias in: every read happens before the single write).
template<int OP, typename S, int N, bool ALIGNED>
__global__ __launch_bounds__(256) void reduce_kernel(const Elem<S, N>* __restrict__ in, Elem<S, N>* __restrict__ out,
                                                     uint64_t count)
{
    using T = Elem<S, N>;
    constexpr int VEC = (ALIGNED && sizeof(T) < 16) ? 16 / (int) sizeof(T) : 1;
    constexpr int UNROLL = 4;
    using P = Pack<T, VEC>;
    __shared__ T wtmp[4];
    __shared__ uint32_t whas[4];
    const uint32_t tid = threadIdx.x;
    T acc = zero_elem<S, N>();
    bool has = false;
    auto fold = [&](const T& v) {
        acc = has ? combine<OP>(acc, v) : v;
        has = true;
    };

    const uint64_t npacks = count / VEC;
    const P* pin = reinterpret_cast<const P*>(in);
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + tid;
    for (; i + (UNROLL - 1) * stride < npacks; i += UNROLL * stride)
    {
        P v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = load_streaming(&pin[i + u * stride]); // read once
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int k = 0; k < VEC; k++) fold(v[u].v[k]);
    }
    for (; i < npacks; i += stride)
    {
        P v = pin[i];
#pragma unroll
        for (int k = 0; k < VEC; k++) fold(v.v[k]);
    }
    if (VEC > 1 && blockIdx.x == 0)
    {
        const uint64_t t = npacks * VEC + tid; // < VEC leftover elements
        if (t < count) fold(in[t]);
    }
    bool rh;
    T r = block_reduce<OP>(acc, has, wtmp, whas, tid, rh);
    if (tid == 0 && rh) out[blockIdx.x] = r;
}

} // namespace glu_hip
