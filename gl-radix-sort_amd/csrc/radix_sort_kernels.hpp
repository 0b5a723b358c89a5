// radix_sort_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the LSD radix sort.
//
// Replaces the device side of glu::RadixSort (reference glu/RadixSort.hpp):
//   K1  radix_count_kernel    <- k_radix_sort_counting_shader   (RadixSort.hpp:11-58)
//   K2  radix_row_scan_kernel <- BlellochScan upsweep/downsweep over the [radix][block] table
//                                (BlellochScan.hpp:13-76, called at RadixSort.hpp:311)
//   K4  radix_scatter_kernel  <- k_radix_sort_reordering_shader (RadixSort.hpp:60-183)
// Same contract (stable counting pass on one digit), different machine mapping:
//   * a fixed grid of persistent workgroups, each owning a contiguous range of tiles, so the digit table is
//     [RADIX][num_blocks] with num_blocks <= a few thousand instead of 16 x N/1024;
//   * per-workgroup digit counts live in registers/LDS, one plain store per (digit, block) -- no global atomics;
//   * stable local ranks from wave64 ballots (match-any on the digit bits) + mbcnt, wave-private LDS counters;
//   * (key, val) staged through LDS in ranked order so that every digit leaves as one contiguous burst.
// The digit width BITS is a template parameter (4 = the reference's pass structure, 8 = half the passes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace glu_hip
{
constexpr int kWave = 64;

template<typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, uint32_t shift, uint32_t mask)
{
    return (uint32_t) (key >> shift) & mask;
}

// Contiguous tile range [first, last) owned by workgroup `b` of `nb`, tiles_total >= nb.
__device__ __forceinline__ void block_tile_range(uint32_t b, uint32_t nb, uint32_t tiles_total, uint32_t& first,
                                                 uint32_t& last)
{
    uint32_t q = tiles_total / nb, r = tiles_total % nb;
    first = b * q + (b < r ? b : r);
    last = first + q + (b < r ? 1u : 0u);
}

__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v, uint32_t lane, uint32_t& total)
{
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1)
    {
        uint32_t t = __shfl_up(incl, off, kWave);
        if (lane >= (uint32_t) off) incl += t;
    }
    total = __shfl(incl, kWave - 1, kWave);
    return incl - v;
}

// ---------------------------------------------------------------------------------------------------------
// K1: per-workgroup digit histogram.  table[d * num_blocks + b] = #keys of block b's range with digit d.
// Reads 1 key per pair (sizeof(KeyT) bytes), writes RADIX counters per workgroup.
// ---------------------------------------------------------------------------------------------------------
template<typename KeyT, int BITS, int THREADS, int TILE>
__global__ __launch_bounds__(THREADS) void radix_count_kernel(const KeyT* __restrict__ keys,
                                                              uint32_t* __restrict__ table, uint32_t n,
                                                              uint32_t shift, uint32_t mask, uint32_t tiles_total)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int WAVES = THREADS / kWave;
    const uint32_t MASK = mask; // <= RADIX - 1 (narrower digits reuse the next wider instantiation)
    __shared__ uint32_t hist[WAVES][RADIX];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();

    uint32_t first, last;
    block_tile_range(blockIdx.x, gridDim.x, tiles_total, first, last);
    const uint64_t begin = (uint64_t) first * TILE;
    uint64_t end = (uint64_t) last * TILE;
    if (end > n) end = n;

    uint32_t* my_hist = hist[wave];

    // every lane of the wave is active when this runs (wave-uniform trip counts below)
    auto tally = [&](uint32_t d) {
        // all-equal digits in the wave (constant / heavily duplicated keys): one add instead of a
        // 64-way same-address LDS atomic
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        if (__ballot(d != d0) == 0)
        {
            if (lane == 0) atomicAdd(&my_hist[d0], 64u);
        }
        else
        {
            atomicAdd(&my_hist[d], 1u);
        }
    };

    constexpr int VEC = 16 / sizeof(KeyT); // 16-byte loads
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    const uint64_t nvec = (end - begin) / VEC;
    const VecT* vkeys = reinterpret_cast<const VecT*>(keys + begin); // begin % TILE == 0 and keys is 16-B aligned
    uint64_t vbase = 0;
    for (; vbase + 2 * THREADS <= nvec; vbase += 2 * THREADS) // block-uniform trip count
    {
        VecT a = vkeys[vbase + tid];
        VecT b = vkeys[vbase + tid + THREADS];
        if constexpr (sizeof(KeyT) == 4)
        {
            tally(digit_of<uint32_t>(a.x, shift, MASK)); tally(digit_of<uint32_t>(a.y, shift, MASK));
            tally(digit_of<uint32_t>(a.z, shift, MASK)); tally(digit_of<uint32_t>(a.w, shift, MASK));
            tally(digit_of<uint32_t>(b.x, shift, MASK)); tally(digit_of<uint32_t>(b.y, shift, MASK));
            tally(digit_of<uint32_t>(b.z, shift, MASK)); tally(digit_of<uint32_t>(b.w, shift, MASK));
        }
        else
        {
            tally(digit_of<uint64_t>(a.x, shift, MASK)); tally(digit_of<uint64_t>(a.y, shift, MASK));
            tally(digit_of<uint64_t>(b.x, shift, MASK)); tally(digit_of<uint64_t>(b.y, shift, MASK));
        }
    }
    // tail (< 2 * THREADS vectors + a partial vector): plain per-key atomics, lanes may be inactive
    for (uint64_t i = begin + vbase * VEC + tid; i < end; i += THREADS)
        atomicAdd(&my_hist[digit_of<KeyT>(keys[i], shift, MASK)], 1u);
    __syncthreads();

    for (int d = tid; d < RADIX; d += THREADS)
    {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) c += hist[w][d];
        table[(size_t) d * gridDim.x + blockIdx.x] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: one workgroup per digit row: in-place exclusive scan of table[d][0..num_blocks), row total to totals[d].
// (The reference scans 16 partitions of nbp2 entries with 2*log2(nbp2) dispatches and keeps the 16 digit
//  totals in a separate buffer that the reorder shader scans itself: RadixSort.hpp:148-152, 311.)
// ---------------------------------------------------------------------------------------------------------
template<int THREADS>
__global__ __launch_bounds__(THREADS) void radix_row_scan_kernel(uint32_t* __restrict__ table,
                                                                 uint32_t* __restrict__ totals, uint32_t num_blocks)
{
    constexpr int WAVES = THREADS / kWave;
    __shared__ uint32_t wave_sums[WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* row = table + (size_t) blockIdx.x * num_blocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_blocks; base += THREADS)
    {
        uint32_t i = base + tid;
        uint32_t v = i < num_blocks ? row[i] : 0;
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(v, lane, wtotal);
        if (lane == 0) wave_sums[wave] = wtotal;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++)
        {
            uint32_t s = wave_sums[w];
            if ((uint32_t) w < wave) woff += s;
            all += s;
        }
        if (i < num_blocks) row[i] = carry + woff + excl;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// ---------------------------------------------------------------------------------------------------------
// K4: stable scatter of (key, val) by one digit.
// ---------------------------------------------------------------------------------------------------------
template<typename KeyT, int BITS, int THREADS, int KPT>
struct ScatterSmem
{
    static constexpr int RADIX = 1 << BITS;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    KeyT keys[TILE];
    uint32_t vals[TILE];
    uint32_t wcnt[WAVES][RADIX]; // wave-private running digit counters -> wave/digit start positions
    uint32_t gdelta[RADIX];      // global index = local position + gdelta[digit]
    uint32_t scan_tmp[WAVES];
};

template<typename KeyT, int BITS, int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(
    const KeyT* __restrict__ src_keys, const uint32_t* __restrict__ src_vals, KeyT* __restrict__ dst_keys,
    uint32_t* __restrict__ dst_vals, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
    uint32_t n, uint32_t shift, uint32_t mask, uint32_t tiles_total)
{
    using Smem = ScatterSmem<KeyT, BITS, THREADS, KPT>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int TILE = Smem::TILE;
    constexpr int WAVE_TILE = kWave * KPT;
    const uint32_t MASK = mask; // <= RADIX - 1
    static_assert(RADIX <= THREADS, "one thread per digit in the offset phase");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = gridDim.x, b = blockIdx.x;

    // ---- prologue: this workgroup's global base for every digit:
    //      exclusive scan of the digit totals (RadixSort.hpp:148-152) + this block's scanned table entry (:176)
    uint32_t digit_base = 0; // valid in threads tid < RADIX
    {
        uint32_t t = tid < RADIX ? totals[tid] : 0;
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(t, lane, wtotal);
        if (lane == 0) s.scan_tmp[wave] = wtotal;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++)
            if ((uint32_t) w < wave) woff += s.scan_tmp[w];
        if (tid < RADIX) digit_base = woff + excl + table[(size_t) tid * nb + b];
    }
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&s.wcnt[0][0])[i] = 0;
    __syncthreads();

    uint32_t first, last;
    block_tile_range(b, nb, tiles_total, first, last);

    for (uint32_t tile = first; tile < last; tile++)
    {
        const uint64_t tile_base = (uint64_t) tile * TILE;
        const uint32_t tile_valid = (n - tile_base) < (uint64_t) TILE ? (uint32_t) (n - tile_base) : (uint32_t) TILE;

        // ---- load: wave-striped (item i of lane l of wave w is element w*WAVE_TILE + i*64 + l), so that
        //      "item-major, then lane" order inside a wave is memory order -> ranks are stable
        KeyT key[KPT];
        uint32_t val[KPT];
        const uint32_t wave_off = wave * WAVE_TILE + lane;
        if (tile_valid == (uint32_t) TILE)
        {
#pragma unroll
            for (int i = 0; i < KPT; i++) key[i] = src_keys[tile_base + wave_off + i * kWave];
#pragma unroll
            for (int i = 0; i < KPT; i++) val[i] = src_vals[tile_base + wave_off + i * kWave];
        }
        else
        {
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                uint32_t p = wave_off + i * kWave;
                bool ok = p < tile_valid;
                key[i] = ok ? src_keys[tile_base + p] : (KeyT) ~(KeyT) 0; // pad: last digit, ranks after all real keys
                val[i] = ok ? src_vals[tile_base + p] : 0u;
            }
        }

        // ---- rank inside the wave: peers = lanes holding the same digit (match-any via one ballot per bit)
        uint32_t rank[KPT];
        uint32_t* my_cnt = s.wcnt[wave];
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            uint64_t peers = ~0ull;
#pragma unroll
            for (int bit = 0; bit < BITS; bit++)
            {
                const bool set = (d >> bit) & 1u;
                const uint64_t m = __ballot(set);
                peers &= set ? m : ~m;
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi((uint32_t) (peers >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((uint32_t) peers, 0u));
            const uint32_t total = (uint32_t) __popcll(peers);
            const uint32_t prev = my_cnt[d];
            rank[i] = prev + lower;
            __builtin_amdgcn_wave_barrier();
            if (lower + 1 == total) my_cnt[d] = prev + total; // highest peer lane publishes the new count
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        // ---- tile offsets: thread d owns digit d
        {
            uint32_t wexcl[WAVES];
            uint32_t dsum = 0;
            if (tid < RADIX)
            {
#pragma unroll
                for (int w = 0; w < WAVES; w++)
                {
                    wexcl[w] = dsum;
                    dsum += s.wcnt[w][tid];
                }
            }
            uint32_t wtotal;
            uint32_t excl = wave_exclusive_sum(dsum, lane, wtotal);
            if (RADIX > kWave)
            {
                if (lane == 0) s.scan_tmp[wave] = wtotal;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < WAVES; w++)
                    if ((uint32_t) w < wave) excl += s.scan_tmp[w];
            }
            if (tid < RADIX)
            {
                // excl = first local position of digit `tid` in the ranked tile
#pragma unroll
                for (int w = 0; w < WAVES; w++) s.wcnt[w][tid] = excl + wexcl[w];
                s.gdelta[tid] = digit_base - excl;
                // pads were counted in the last digit only; they never reach memory and the last digit's base
                // is not used after the final tile
                digit_base += dsum;
            }
        }
        __syncthreads();

        // ---- stage in ranked order
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            const uint32_t pos = my_cnt[d] + rank[i];
            s.keys[pos] = key[i];
            s.vals[pos] = val[i];
        }
        __syncthreads();

        // ---- write out: consecutive threads -> consecutive local positions -> (per digit) consecutive addresses
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t p = i * THREADS + tid;
            if (p < tile_valid)
            {
                const KeyT k = s.keys[p];
                const uint32_t g = p + s.gdelta[digit_of<KeyT>(k, shift, MASK)];
                dst_keys[g] = k;
                dst_vals[g] = s.vals[p];
            }
        }
        for (int i = tid; i < WAVES * RADIX; i += THREADS) (&s.wcnt[0][0])[i] = 0;
        __syncthreads();
    }
}

} // namespace glu_hip
