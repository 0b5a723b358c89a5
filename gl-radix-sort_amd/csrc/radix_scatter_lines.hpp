// radix_scatter_lines.hpp -- the large-input scatter pass of the LSD radix sort, written around whole 128-byte lines.
//
// Replaces k_radix_sort_reordering_shader (reference glu/RadixSort.hpp:60-183) on inputs large enough for one persistent
// 1024-thread workgroup per CU, in 16-byte aligned arrays (the other cases keep radix_scatter_kernel of
// radix_sort_kernels.hpp).
// Same contract: a stable counting pass on one digit, dst = digit base + block offset + local rank (RadixSort.hpp:174-177).
//
// Why a second kernel.  On MI355X what the memory system delivers for a streaming read + scattered write mix depends on
// the write chunks alone (tools/pattern_bench.hip, 2^28 pairs, 256 destination regions per workgroup):
//     64-B chunks 3.5-3.8 TB/s | 192-B chunks at 64-B alignment 3.8 TB/s | 128-B-aligned chunks of 128 B and up 5.0-5.1 TB/s
// (5.1-5.4 TB/s is the plain copy).  radix_scatter_kernel writes 64-byte blocks (its carry granule) and sat exactly on the
// 3.8 TB/s line.  Here every store to memory is a whole, aligned 128-byte line of one digit's output, written by 8 lanes
// with one 16-byte store each:
//   * per digit a carry of up to 31 elements (the part of the digit's output past its last 128-byte boundary) stays in
//     LDS between tiles; a tile appends its run; the full lines of (carry ++ run) are written, the rest is the new carry;
//   * the write-out walks LINES, not elements: a packed block scan gives every digit its first ranked position (low half)
//     and its first line (high half) at once; a line -> digit table makes a quad of 4 elements = 1 lane-store;
//   * the only partial lines are at the two ends of a workgroup's range of a digit (element-wise stores).
// The tile loop is a software pipeline: the next tile's keys are loaded while the current one is ranked, its values while the
// current one is staged, and the next tile is ranked (VALU-bound, wave-private state) beside the LDS-bound staging and tail
// copy of the current one; 5 barriers per tile, none at its end.  The line stores are non-temporal (the count kernel of the
// next pass reads them once: it runs 25 % faster when they do not sit dirty in L2 / Infinity Cache).
#pragma once

#include "radix_sort_kernels.hpp"

namespace glu_hip
{
// Elements per carry granule = keys per 128-byte line: 32 for 4-byte keys (values then fill whole lines too), 16 for
// 8-byte keys (whole lines of keys, 64-byte halves of value lines: a 32-element granule would need 96 KiB of carry).
template<typename KeyT>
constexpr int line_elems() { return 128 / (int) sizeof(KeyT); }

template<typename KeyT, int BITS, int THREADS, int KPT, bool VALS = true>
struct LineSmem
{
    static constexpr int RADIX = 1 << BITS;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    static constexpr int LINE = line_elems<KeyT>();
    // A digit's carry row holds LINE slots.  Rows of 4-byte keys are LINE + 1 slots apart: with rows of exactly 32 x 8 B = 256 B
    // every digit's slot s lies in the same LDS bank column, and the two phases whose lanes walk (digit, quad of slots) pairs
    // -- the tail copy's writes (16 lanes = 2 digits x 8 quads onto 4 columns: 4-way on every store) and the first lines'
    // reads -- serialise; one slot of padding rotates the columns by the digit.  (8-byte keys: no LDS left for it.)
    static constexpr int CSTRIDE = LINE + (sizeof(KeyT) == 4 ? 1 : 0);
    static constexpr int CARRY = RADIX * CSTRIDE;
    static constexpr int MAXLINES = (TILE + RADIX * (LINE - 1)) / LINE + 2;
    static_assert(TILE < 65536 && MAXLINES < 65536, "ranked positions and line numbers share one 32-bit scan word");
    static_assert(RADIX <= 256, "the line table holds digits as bytes");
    PairArray<KeyT, TILE + CARRY, VALS> buf; // [0, TILE): the tile in ranked order;  [TILE + d * CSTRIDE, + LINE): carry of digit d
    // 16-bit counters (a wave ranks at most 64 * KPT elements of a tile, positions stay below TILE < 65536): half the LDS
    // of 32-bit ones, which the tile gets.  Rows of RADIX + 4 halfwords: the scan's four rows per thread land 8 banks apart.
    static constexpr int WCNT_STRIDE = RADIX + 4;
    uint16_t wcnt[WAVES][WCNT_STRIDE]; // wave-private running digit counters -> first ranked position of (wave, digit)
    uint4 dinfo[RADIX];  // .x global index of the digit's first line this tile (= of its carried elements), .y ranked position
                         // of combined element 0 (= first ranked position - carried count), .z first line | carried << 16,
                         // .w first global index of the digit that this workgroup owns
    uint32_t tail[RADIX]; // new carry: slots [lo, hi) <- ranked positions (p + slot) mod 2^16; p | lo << 16 | hi << 24
    // 8-byte keys with values: a line is 16 elements = a whole 128-byte line of keys but HALF a line of values.  When the last
    // line a digit emits in a tile is the first half of a value line (its second half follows a tile period later, when the
    // L2 has long dropped the half-written line: fill reads, traffic 1.13 x algorithmic), its 16 values wait here instead
    // and leave together with the second half.
    static constexpr bool SHADOW = sizeof(KeyT) == 8 && VALS;
    uint32_t shadow[SHADOW ? RADIX * LINE : 1];
    uint32_t shadow_out[SHADOW ? RADIX : 1]; // global index of the shadowed line that leaves in this tile's scan phase, or ~0
    uint8_t ltab[(MAXLINES + 3) & ~3]; // digit of every line written this tile
    uint32_t scan_tmp[WAVES];
    uint32_t total_lines;
};

// ABLATE (tuning builds, wrong results): 4 = nothing is written out.  RANK_SPLIT: how many of a tile's KPT items are
// ranked right after the staging of the tile before (the rest after that tile's tail copy).  STAGGER: SIMD partner waves
// take tail copy and ranking in opposite order.  NT_STORES: non-temporal line stores (what the library runs).  PRIO:
// tuning only (s_setprio around the LDS-bound phases: measured, no effect).  STAMPS: s_memtime per phase of the first and
// the last wave into stamps[0..15] (tools/scatter_bench.hip).
//
// BEHIND_ATTEMPT changes nothing but the kernel's name: the ordinary passes enqueued behind an attempt to end the sort in LDS
// (radix_lds_finish.hpp), which return at once when the attempt was accepted.
// RANK_ATOMIC (measured in round 3, NOT used by the library; tools/scatter_bench.hip SB_RA=1): an item's rank inside its wave
// from ONE returning LDS atomic add on the wave's counter row (two 16-bit counters per word) instead of eight ballots.  A
// stable pass needs the lanes of one instruction that add to the same counter served in lane order: gfx950 does that
// (tools/lds_atomic_order.hip: 0 of 5.2 x 10^9 items out of order), the ISA document does not promise it.  It takes a fifth
// off the kernel's compute (17.1 k -> 13.5 k cycles per tile with 48 workgroups) and nothing off its time on the whole chip
// (0.75-0.77 ms on a fast device either way, 0.912 -> 0.89 ms on a slow one, no difference inside the sort on five devices:
// DESIGN.md section 4.2) -- the kernel is not bound by its instruction issue, so the undocumented order is not relied upon.
//
// SEG (segmented passes: the local sort of the sharded sort, glu_dist_impl.hpp): the workgroup runs the whole pass body once
// per SUB-BLOCK of its list [seg_first[b], seg_first[b + 1]): element range `ranges[i]` of the source arrays, counted on its
// own (radix_seg_count_kernel), with `table[i * RADIX + d]` = the ABSOLUTE destination index of the sub-block's first
// element with digit d (radix_seg_scan_kernel: segment start + digits below d in the segment + the segment's earlier
// sub-blocks).  Everything else -- ranking, line carry, whole-line stores, element-wise stores at the two ends of a
// (sub-block, digit) range -- is the pass as above; `totals`, `plan`, `share` are not used.
template<typename KeyT, int BITS, int THREADS, int KPT, bool XF = false, bool VALS = true, int ABLATE = 0, bool STAMPS = false,
         int RANK_SPLIT = (KPT + 2) / 3, bool STAGGER = true, bool NT_STORES = false, int PRIO = 0, bool SEG = false,
         bool RANK_ATOMIC = false, bool BEHIND_ATTEMPT = false>
__global__ __launch_bounds__(THREADS) void radix_scatter_lines_kernel(
    const KeyT* __restrict__ keys_a, const uint32_t* __restrict__ vals_a, KeyT* __restrict__ keys_b,
    uint32_t* __restrict__ vals_b, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals, uint32_t n,
    uint32_t shift, uint32_t mask, uint32_t tiles_total, unsigned long long* stamps = nullptr, uint32_t xform = 0,
    PassPlan* plan = nullptr, uint32_t pass = 0, const uint2* __restrict__ ranges = nullptr, uint32_t share = 0,
    const uint32_t* __restrict__ seg_first = nullptr, const uint32_t* gate = nullptr, uint32_t gate_cap = 0, uint32_t gate_mode = 0,
    const uint32_t* seg_shares = nullptr)
{
    // (SEG: a pass of a segmented sort that may end in LDS runs or returns by the longest run, radix_seg_passes.hpp)
    if (SEG && gate_mode != 0 && ((*gate <= gate_cap) != (gate_mode == 1))) return; // (kernel-uniform)
    const KeyT* __restrict__ src_keys = keys_a;
    const uint32_t* __restrict__ src_vals = vals_a;
    KeyT* __restrict__ dst_keys = keys_b;
    uint32_t* __restrict__ dst_vals = vals_b;
    if (SEG && plan && ((plan->flip[pass] ^ share) & 1u))
    {
        // (the long runs of a whole-key sort that ends in LDS: which pair of arrays holds the data is known on the device only;
        // `share`, otherwise unused by segmented passes, is 1 for the pass that goes back)
        src_keys = keys_b;
        src_vals = vals_b;
        dst_keys = const_cast<KeyT*>(keys_a);
        dst_vals = const_cast<uint32_t*>(vals_a);
    }
    if (!SEG && plan)
    {
        shift -= plan->shift_down[pass]; // (PassPlan::top_bit)
        const uint32_t flip = pass > 0 ? plan->flip[pass] : 0u;
        const uint32_t skip = plan->skip[pass];
        if (blockIdx.x == 0 && threadIdx.x == 0) plan->flip[pass + 1] = flip ^ (skip ? 0u : 1u);
        if (skip) return; // identity pass (kernel-uniform)
        if (flip)
        {
            src_keys = keys_b;
            src_vals = vals_b;
            dst_keys = const_cast<KeyT*>(keys_a);
            dst_vals = const_cast<uint32_t*>(vals_a);
        }
    }
    using Smem = LineSmem<KeyT, BITS, THREADS, KPT, VALS>;
    constexpr int RADIX = Smem::RADIX;
    constexpr int WAVES = Smem::WAVES;
    constexpr int TILE = Smem::TILE;
    constexpr int WAVE_TILE = kWave * KPT;
    constexpr int SCAN_WAVES = (RADIX + kWave - 1) / kWave; // the waves that hold one digit per lane in the scan phase
    constexpr uint32_t LINE = Smem::LINE;
    static_assert(RADIX <= THREADS, "one scan thread per digit");
    const uint32_t MASK = mask; // <= RADIX - 1

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const KeyCodec<KeyT, XF> codec_in(xform & 3u), codec_out((xform >> 2) & 3u);

    // thread d < RADIX owns digit d in the scan phase and keeps the digit's running state in registers:
    const uint32_t sd = tid;
    const bool digit_owner = tid < (uint32_t) RADIX;
    uint32_t digit_base = 0;  // global index of the digit's next element
    uint32_t carry_start = 0; // 32-aligned global index of the digit's first carried element; carried = digit_base - carry_start
    uint32_t owned_from = 0;  // first global index of the digit inside this workgroup's range
    constexpr bool SHADOW = Smem::SHADOW;
    bool shadow_valid = false; // SHADOW: s.shadow[sd] holds the values of the line at global index shadow_pos (keys written already)
    uint32_t shadow_pos = 0;

    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int slot) {
        if (STAMPS)
        {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            acc[slot] += t - tprev;
            tprev = t;
        }
    };

    const uint32_t wave_off = wave * WAVE_TILE + lane; // wave-striped: item i of lane l of wave w = element w*64*KPT + i*64 + l
    // (readfirstlane: the values are workgroup-uniform; it keeps them and everything derived from them in scalar registers)
    // (SEG with seg_shares: more shares than workgroups, workgroup b takes the shares b, b + nb, ... -- radix_seg_count_kernel)
    // (do ... while (SEG && ...): the unsegmented kernel keeps no loop around its body -- a loop the compiler could not see through
    // cost every instantiation 36 bytes of scratch per lane)
    const uint32_t seg_nshares = SEG ? (seg_shares ? (uint32_t) __builtin_amdgcn_readfirstlane((int) *seg_shares) : nb) : 0u;
    uint32_t seg_sh = b;
    do
    {
    const uint32_t sb_first = SEG ? (uint32_t) __builtin_amdgcn_readfirstlane((int) seg_first[seg_sh]) : 0u;
    const uint32_t sb_last = SEG ? (uint32_t) __builtin_amdgcn_readfirstlane((int) seg_first[seg_sh + 1]) : 1u;
    for (uint32_t sb = sb_first; sb < sb_last; sb++)
    {
    // ---- prologue: exclusive scan of the digit totals (RadixSort.hpp:148-152) + this block's scanned table entry (:176)
    if (SEG)
    {
        if (digit_owner)
        {
            digit_base = table[(size_t) sb * RADIX + sd]; // absolute: the segmented scan has added everything
            owned_from = digit_base;
            carry_start = digit_base & ~(LINE - 1);
        }
    }
    else
    {
        const uint32_t t = digit_owner ? totals[sd] : 0u;
        uint32_t wtotal;
        uint32_t excl = wave_exclusive_sum(t, lane, wtotal);
        if (lane == 0) s.scan_tmp[wave] = wtotal;
        __syncthreads();
        excl += sum_of_preceding_waves(s.scan_tmp, WAVES, wave, lane);
        if (digit_owner)
        {
            digit_base = excl + table[(size_t) sd * nb + b];
            owned_from = digit_base;
            carry_start = digit_base & ~(LINE - 1); // the slots below digit_base are never written (another workgroup's)
        }
    }
    static_assert(Smem::WCNT_STRIDE % 2 == 0, "counter rows are zeroed as 32-bit words");
    for (int i = tid; i < WAVES * Smem::WCNT_STRIDE / 2; i += THREADS) reinterpret_cast<uint32_t*>(&s.wcnt[0][0])[i] = 0;
    __syncthreads();

    // The workgroup's element range [r0, r1), cut into tiles from r0: an equal share of the elements (or whole tiles of
    // the array: tuning harness), or -- the follower of a pair of passes (radix_pair_passes.hpp) -- a run of whole units of
    // the pass before, which starts and ends at any element, or -- SEG -- the sub-block's range.  Only the last tile of a
    // range can be partial.
    // (32-bit element indices: the library refuses counts beyond 0xFFFF0000, and a prefetch reaches at most two tiles further;
    // per-lane 64-bit indices cost two registers each, which this kernel -- at its 128-register limit -- does not have)
    uint32_t r0, r1;
    if (SEG)
    {
        const uint2 r = ranges[sb];
        r0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) r.x);
        r1 = (uint32_t) __builtin_amdgcn_readfirstlane((int) r.y);
    }
    else if (ranges && !plan->pair_fallback[pass])
    {
        const uint2 r = ranges[b];
        r0 = r.x;
        r1 = r.y;
    }
    else
    {
        uint64_t b0, b1;
        block_range(b, nb, tiles_total, TILE, n, share, b0, b1); // whole tiles (share == 0) or an equal share of the elements
        r0 = (uint32_t) b0;
        r1 = (uint32_t) b1;
    }
    const uint32_t first = 0, last = (uint32_t) (((uint64_t) (r1 - r0) + (uint64_t) TILE - 1) / (uint64_t) TILE);
    const uint32_t last_tile_of_range = last;

    KeyT key[KPT], nkey[KPT];
    uint32_t val[KPT];
    uint32_t rank[KPT];
    uint16_t* const my_cnt = s.wcnt[wave];
    // guarded loads of a partial tile: positions past the end of the array read as pads (highest digit, after all keys)
    auto load_tile_guarded = [&](uint32_t t) {
        const uint32_t base = r0 + t * (uint32_t) TILE;
        const uint32_t left = r1 - base;
#pragma unroll
        for (int i = 0; i < KPT; i++)
        {
            const uint32_t p = wave_off + i * kWave;
            key[i] = p < left ? codec_in.encode(src_keys[base + p]) : (KeyT) ~(KeyT) 0;
            if (VALS) val[i] = p < left ? src_vals[base + p] : 0u;
        }
    };
    // full tile t, or the first tile of the array when t is not a full tile of this range (a harmless prefetch of
    // something that is not used: the loops that prefetch carry no branch)
    auto prefetch_base = [&](uint32_t t) -> uint32_t {
        const uint32_t tb = r0 + t * (uint32_t) TILE;
        const bool ok = t < last_tile_of_range && r1 - tb >= (uint32_t) TILE;
        return (ok ? tb : 0u) + wave_off;
    };
    // Ranks items [I0, I1) of the tile in key[] inside the wave (match-any on the digit bits with one ballot per bit,
    // wave-private running counters) and loads the same items of the tile after it into nkey[].  Touches nothing but the
    // wave's own counter row, so it needs no workgroup barrier against the phases of the tile before.
    // (Tried: ballots for all items first, then the counters with one read and one leader-only atomic add per item, so
    // that the ballots carry no LDS dependency: 6 % slower, the adds and their exec-mask branches cost more than the
    // dependency they remove.)
    auto rank_items = [&](auto i0, auto i1, uint32_t pf_base) {
#pragma unroll
        for (int i = decltype(i0)::value; i < decltype(i1)::value; i++)
        {
            nkey[i] = src_keys[pf_base + i * kWave]; // (non-temporal loads here: 2 % slower inside the sort, same-box A/B)
            const uint32_t d = digit_of<KeyT>(key[i], shift, MASK);
            if constexpr (RANK_ATOMIC)
            {
                const uint32_t half = (d & 1u) * 16u;
                const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(my_cnt) + (d >> 1), 1u << half); // ds_add_rtn_u32
                rank[i] = (old >> half) & 0xFFFFu; // the digit's count before this instruction + the lower lanes with it
                continue;
            }
            uint16_t* const cnt = my_cnt + d;
            const uint32_t prev = *cnt; // issued first: its LDS latency hides under the ballots below
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (int bit = 0; bit < BITS; bit++)
            {
                int32_t sel; // asm: keeps bit 0 from being rewritten as -(d & 1) and a compare chain
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(d), "n"(bit));
                const uint64_t m = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t) m, (uint32_t) sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t) (m >> 32), (uint32_t) sel, 0x90);
            }
            // v_mbcnt and v_bcnt add into an accumulator operand: rank and new count come out of two instructions each
            rank[i] = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, prev)); // prev + lower peers
            asm volatile("" : "+v"(rank[i])); // keep the rank (1 register), not the two peer masks, live
            uint32_t new_count; // prev + number of peers; asm: hipcc does not fold the additions into v_bcnt's accumulator
            asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(new_count) : "v"(plo), "v"(prev));
            asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(new_count) : "v"(phi), "v"(new_count));
            *cnt = (uint16_t) new_count; // every peer stores the same new count (no leader election)
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using ISplit = std::integral_constant<int, RANK_SPLIT>;
    using IEnd = std::integral_constant<int, KPT>;
    constexpr uint32_t CSTRIDE = Smem::CSTRIDE;
    // ---- new carry: the elements of every digit past its last full line move to the digit's carry slots.  Work items are
    //      (digit, quad of slots).
    constexpr int TAIL_ITEMS = RADIX * (int) (LINE / 4);
    auto copy_tails = [&]() {
        if (PRIO & 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int it = 0; it < (TAIL_ITEMS + THREADS - 1) / THREADS; it++)
        {
            const uint32_t item = it * THREADS + tid;
            if (TAIL_ITEMS % THREADS == 0 || item < (uint32_t) TAIL_ITEMS)
            {
                const uint32_t d = item / (LINE / 4), s0 = (item % (LINE / 4)) * 4;
                const uint32_t t = s.tail[d];
                const uint32_t from = t & 0xFFFFu, lo = (t >> 16) & 0xFFu, hi = t >> 24;
                KeyT k[4];
                uint32_t v[4];
                // slots below lo read the element of slot lo (a valid position), slots from hi up read past the run
                // (inside the buffer): neither is written.  (`from` is the ranked position of slot 0 modulo 2^16: it lies
                // before the digit's run -- "negative" when the run is short -- and only from + slot, slot >= lo, is a position)
#pragma unroll
                for (int e = 0; e < 4; e++) s.buf.get((from + (s0 + e > lo ? s0 + e : lo)) & 0xFFFFu, k[e], v[e]);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (s0 + e >= lo && s0 + e < hi) s.buf.put((uint32_t) TILE + d * CSTRIDE + s0 + e, k[e], v[e]);
            }
        }
        if (PRIO & 2) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue of the software pipeline: the first tile is loaded and ranked here; every later tile is loaded and
    //      ranked under the phases of the tile before it
    if (first < last)
    {
        if (r1 - r0 >= (uint32_t) TILE)
        {
            const uint32_t base = r0 + wave_off;
#pragma unroll
            for (int i = 0; i < KPT; i++) key[i] = codec_in.encode(src_keys[base + i * kWave]);
            if (VALS)
            {
#pragma unroll
                for (int i = 0; i < KPT; i++) val[i] = src_vals[base + i * kWave];
            }
        }
        else
            load_tile_guarded(first);
        rank_items(I0(), IEnd(), prefetch_base(first + 1));
    }

    for (uint32_t tile = first; tile < last; tile++)
    {
        const uint32_t tile_base = r0 + tile * (uint32_t) TILE;
        const uint32_t rem = r1 - tile_base;
        const uint32_t tile_valid = rem >= (uint32_t) TILE ? (uint32_t) TILE : rem;
        const bool has_next = tile + 1 < last;
        const bool next_full = has_next && r1 - (tile_base + (uint32_t) TILE) >= (uint32_t) TILE;
        if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
        __syncthreads(); // every wave has ranked this tile
        stamp(0);

        // ---- one packed block scan over the digits: low half = elements (first ranked position of the digit), high half =
        //      full lines of the digit.  Only the RADIX / 64 waves that hold a digit per lane work here (the whole kernel is
        //      bound by vector-instruction issue: what the other waves do not execute is time the SIMDs get back); each of
        //      their threads reads the digit's WAVES counters, and writes back WAVES prefixes after the scan.
        {
            uint32_t c[WAVES];
            uint32_t n_d = 0, c_d = 0, nl_d = 0, excl = 0;
            if (wave < SCAN_WAVES) // wave-uniform
            {
                uint32_t n_raw = 0;
                if (digit_owner)
                {
#pragma unroll
                    for (int w = 0; w < WAVES; w++) c[w] = s.wcnt[w][sd];
#pragma unroll
                    for (int w = 0; w < WAVES; w++) n_raw += c[w];
                    // pads of the (partial) last tile were ranked at the end of the highest used digit: not elements
                    n_d = n_raw - (sd == MASK ? (uint32_t) TILE - tile_valid : 0u);
                    c_d = digit_base - carry_start;
                    nl_d = (c_d + n_d) / LINE;
                    // (SHADOW) the first line this tile is the second half of the value line whose first half waits in the shadow
                    if (SHADOW) s.shadow_out[sd] = shadow_valid && nl_d > 0 ? shadow_pos : 0xFFFFFFFFu;
                }
                uint32_t wtotal;
                excl = wave_exclusive_sum(n_raw | (nl_d << 16), lane, wtotal);
                if (lane == 0) s.scan_tmp[wave] = wtotal;
            }
            __syncthreads();
            if (SHADOW && wave >= SCAN_WAVES)
            {
                // the waves that hold no digit in this phase send the waiting first halves off: 4 lanes x 16 bytes per half
                // line in one store instruction (the second half follows in this tile's write-out, microseconds later).
                // (One thread per digit storing its 64 bytes with four instructions made the pass 30 % slower: every
                // instruction then writes 16 bytes each of 64 different lines.)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                constexpr uint32_t TT = THREADS - SCAN_WAVES * kWave, ITEMS = RADIX * (LINE / 4);
                for (uint32_t item = tid - SCAN_WAVES * kWave; item < ITEMS; item += TT)
                {
                    const uint32_t d = item / (LINE / 4), q = item % (LINE / 4);
                    const uint32_t out = s.shadow_out[d];
                    if (out != 0xFFFFFFFFu)
                    {
                        const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(&s.shadow[d * LINE + q * 4]);
                        if (NT_STORES)
                            __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t*>(dst_vals + out + 4 * q));
                        else
                            *reinterpret_cast<u32x4_t*>(dst_vals + out + 4 * q) = vv;
                    }
                }
            }
            if (wave < SCAN_WAVES)
            {
                if (SCAN_WAVES > 1) excl += sum_of_preceding_waves(s.scan_tmp, SCAN_WAVES, wave, lane);
                const uint32_t pos = excl & 0xFFFFu, line0 = excl >> 16;
                if (digit_owner)
                {
                    uint32_t running = pos;
#pragma unroll
                    for (int w = 0; w < WAVES; w++)
                    {
                        s.wcnt[w][sd] = (uint16_t) running;
                        running += c[w];
                    }
                    const uint32_t m = c_d + n_d, c_new = m & (LINE - 1), k = n_d < c_new ? n_d : c_new;
                    uint32_t orphan_line = 0x3FFu; // SHADOW: the line (index in this tile's line table) whose values go to s.shadow
                    if (SHADOW)
                    {
                        static_assert(!SHADOW || (Smem::MAXLINES < 0x3FF && LINE == 16), "line numbers in 10 bits, carried count in 4");
                        if (shadow_valid && nl_d > 0) shadow_valid = false; // (its values are leaving: the other waves, above)
                        if (nl_d > 0)
                        {
                            const uint32_t last_pos = carry_start + (nl_d - 1) * LINE; // global index of the last line emitted
                            if (((last_pos / LINE) & 1u) == 0u && last_pos >= owned_from)
                            {
                                orphan_line = line0 + nl_d - 1;
                                shadow_valid = true;
                                shadow_pos = last_pos;
                            }
                        }
                    }
                    s.dinfo[sd] = make_uint4(carry_start, pos - c_d, line0 | (c_d << 16) | (SHADOW ? orphan_line << 20 : 0u), owned_from);
                    s.tail[sd] = ((pos + n_d - c_new) & 0xFFFFu) | ((c_new - k) << 16) | (c_new << 24);
                    digit_base += n_d;
                    carry_start += nl_d * LINE;
                    if (tid == (uint32_t) RADIX - 1) s.total_lines = line0 + nl_d;
                }
                // line -> digit table.  Every lane writes the first lines of its digit itself (one or two for uniform keys, 24
                // per digit with 16 digit values: the lanes loop in parallel); what a digit has beyond kOwnLines lines (a digit
                // that holds a large part of the tile: at most TILE / 32 / kOwnLines of them) the wave fills together.
                constexpr uint32_t kOwnLines = 32;
                if (digit_owner)
                    for (uint32_t j = 0; j < (nl_d < kOwnLines ? nl_d : kOwnLines); j++) s.ltab[line0 + j] = (uint8_t) sd;
                uint64_t big = __ballot(digit_owner && nl_d > kOwnLines);
                while (big)
                {
                    const int l = __builtin_ctzll(big);
                    big &= big - 1;
                    const uint32_t bd = (uint32_t) __builtin_amdgcn_readlane((int) sd, l);
                    const uint32_t b0 = (uint32_t) __builtin_amdgcn_readlane((int) line0, l);
                    const uint32_t bn = (uint32_t) __builtin_amdgcn_readlane((int) nl_d, l);
                    for (uint32_t j = kOwnLines + lane; j < bn; j += kWave) s.ltab[b0 + j] = (uint8_t) bd;
                }
            }
        }
        __syncthreads();
        stamp(1);

        // ---- stage (key, val) at the ranked position; the value register just staged takes the next tile's value
        {
            const uint32_t next_base = prefetch_base(tile + 1);
            if (PRIO & 1) __builtin_amdgcn_s_setprio(2); // tuning: LDS-bound phases ahead of the ranking of other waves
#pragma unroll
            for (int i = 0; i < KPT; i++)
            {
                const uint32_t pos = rank[i] + my_cnt[digit_of<KeyT>(key[i], shift, MASK)];
                s.buf.put(pos, key[i], VALS ? val[i] : 0u);
                if (VALS) val[i] = src_vals[next_base + i * kWave];
            }
            if (PRIO & 1) __builtin_amdgcn_s_setprio(0);
        }
        // ---- this wave is done with its counter row: zero it and start ranking the next tile (keys prefetched into nkey
        //      while this tile was ranked; a partial next tile, the last of the array, takes guarded loads now).  The
        //      ranking is VALU work on wave-private state: no barrier separates it from this tile's later phases, the
        //      rest of it follows the tail copy below.
        for (int i = lane; i < Smem::WCNT_STRIDE / 2; i += kWave) reinterpret_cast<uint32_t*>(my_cnt)[i] = 0;
        const uint32_t pf_base = prefetch_base(tile + 2);
        if (has_next)
        {
            if (next_full)
            {
#pragma unroll
                for (int i = 0; i < KPT; i++) key[i] = codec_in.encode(nkey[i]);
            }
            else
                load_tile_guarded(tile + 1);
            rank_items(I0(), ISplit(), pf_base);
        }
        stamp(2);
        __syncthreads();
        stamp(3);

        // ---- write the full lines: quad = 4 consecutive elements of one line = one 16-byte store per array
        auto write_out = [&]() {
        if (ABLATE < 4)
        {
            const uint32_t quads = s.total_lines * (LINE / 4);
            // Quads per thread and sweep: a tile of uniform keys has about 2.5 per thread, so the waves whose third quad would
            // lie past the end (wave-uniform test) take two (the kernel is bound by instruction issue: no surplus work).
            auto write_lines = [&](auto qb, uint32_t q_begin) {
                constexpr int QB = decltype(qb)::value;
            for (uint32_t q_first = q_begin; q_first < quads; q_first += QB * THREADS)
            {
                uint32_t g0[QB], from_run[QB], from_carry[QB], owned[QB];
                int in_carry[QB];
                int to_shadow[SHADOW ? QB : 1]; // SHADOW: >= 0: the quad's values go to these shadow slots, not to memory
#pragma unroll
                for (int u = 0; u < QB; u++)
                {
                    uint32_t qi = q_first + u * THREADS;
                    qi = qi < quads ? qi : q_first; // a surplus quad reads what the first one reads and stores nothing
                    const uint32_t l = qi / (LINE / 4), sub = qi % (LINE / 4);
                    const uint32_t d = s.ltab[l];
                    const uint4 info = s.dinfo[d];
                    const uint32_t line0 = info.z & 0xFFFFu, carried = SHADOW ? (info.z >> 16) & 0xFu : info.z >> 16;
                    if (SHADOW) to_shadow[u] = l == ((info.z >> 20) & 0x3FFu) ? (int) (d * LINE + sub * 4) : -1;
                    const uint32_t q0 = (l - line0) * LINE + sub * 4; // index in (carry ++ run) of the quad's first element
                    g0[u] = info.x + q0;                               // its global index (a multiple of 4)
                    from_run[u] = info.y + q0;                         // ranked position, were it an element of the run
                    from_carry[u] = (uint32_t) TILE + d * CSTRIDE + q0;
                    in_carry[u] = (int) carried - (int) q0;            // elements e < in_carry of the quad are carried ones
                    owned[u] = info.w;
                }
#pragma unroll
                for (int u = 0; u < QB; u++)
                {
                    KeyT k[4];
                    uint32_t v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) s.buf.get((e < in_carry[u] ? from_carry[u] : from_run[u]) + e, k[e], v[e]);
#pragma unroll
                    for (int e = 0; e < 4; e++) k[e] = codec_out.decode(k[e]);
                    const bool real = u == 0 || q_first + u * THREADS < quads;
                    if (real && g0[u] >= owned[u] && (ABLATE == 0 || g0[u] + 3 < n))
                    {
                        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                        if constexpr (sizeof(KeyT) == 4)
                        {
                            u32x4_t kk = {(uint32_t) k[0], (uint32_t) k[1], (uint32_t) k[2], (uint32_t) k[3]};
                            if (NT_STORES)
                                __builtin_nontemporal_store(kk, reinterpret_cast<u32x4_t*>(dst_keys + g0[u]));
                            else
                                *reinterpret_cast<u32x4_t*>(dst_keys + g0[u]) = kk;
                        }
                        else
                        {
                            typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
                            u64x2_t k01 = {(uint64_t) k[0], (uint64_t) k[1]}, k23 = {(uint64_t) k[2], (uint64_t) k[3]};
                            if (NT_STORES)
                            {
                                __builtin_nontemporal_store(k01, reinterpret_cast<u64x2_t*>(dst_keys + g0[u]));
                                __builtin_nontemporal_store(k23, reinterpret_cast<u64x2_t*>(dst_keys + g0[u] + 2));
                            }
                            else
                            {
                                *reinterpret_cast<u64x2_t*>(dst_keys + g0[u]) = k01;
                                *reinterpret_cast<u64x2_t*>(dst_keys + g0[u] + 2) = k23;
                            }
                        }
                        if (VALS)
                        {
                            u32x4_t vv = {v[0], v[1], v[2], v[3]};
                            if (SHADOW && to_shadow[u] >= 0)
                                *reinterpret_cast<u32x4_t*>(&s.shadow[to_shadow[u]]) = vv; // (a fully owned line: see the scan phase)
                            else if (NT_STORES)
                                __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t*>(dst_vals + g0[u]));
                            else
                                *reinterpret_cast<u32x4_t*>(dst_vals + g0[u]) = vv;
                        }
                    }
                    else if (real && ABLATE == 0)
                    {
                        // first line of the digit in this workgroup's range: the slots below owned_from belong to the
                        // workgroup before (it writes them when it flushes its carry)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (g0[u] + e >= owned[u])
                            {
                                dst_keys[g0[u] + e] = k[e];
                                if (VALS) dst_vals[g0[u] + e] = v[e];
                            }
                    }
                }
            }
            };
            if (__builtin_amdgcn_readfirstlane((int) tid) + 2 * THREADS < (int) quads)
                write_lines(std::integral_constant<int, 3>(), tid);
            else
                write_lines(std::integral_constant<int, 2>(), tid);
        }
        };
        write_out();
        stamp(4);
        __syncthreads(); // the carried elements have been read: their slots may be rewritten
        stamp(5);
        // ---- tail copy (LDS-bound) and the rest of the next tile's ranking (VALU-bound, wave-private state: nobody else
        //      touches this wave's counter row between this tile's staging and the scan of the next tile, so the tile
        //      ends without a barrier).  The two are independent: on every SIMD two waves copy first and two rank first, so
        //      that the LDS and the vector ALUs are busy at the same time (waves run in lockstep behind a barrier; the
        //      second four of every eight waves are the SIMD partners of the first four).
        if (STAGGER && (wave & 4u))
        {
            if (has_next) rank_items(ISplit(), IEnd(), pf_base);
            copy_tails();
        }
        else
        {
            copy_tails();
            if (has_next) rank_items(ISplit(), IEnd(), pf_base);
        }
        stamp(6);
    }

    // ---- what is still carried at the end of this workgroup's range: the (partial) last line of every digit
    if (digit_owner) s.dinfo[sd] = make_uint4(carry_start, digit_base - carry_start, 0u, owned_from);
    __syncthreads();
    if (SHADOW && digit_owner && shadow_valid)
    {
        // a first half of a value line whose second half belongs to the workgroup after this one (or does not exist)
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t* sh = reinterpret_cast<const u32x4_t*>(&s.shadow[sd * LINE]);
#pragma unroll
        for (int q = 0; q < (int) LINE / 4; q++) *reinterpret_cast<u32x4_t*>(dst_vals + shadow_pos + 4 * q) = sh[q];
        shadow_valid = false;
    }
    if (ABLATE < 4)
    {
        for (uint32_t e = tid; e < (uint32_t) RADIX * LINE; e += THREADS)
        {
            const uint32_t d = e / LINE, slot = e % LINE;
            const uint4 info = s.dinfo[d];
            const uint32_t g = info.x + slot;
            if (slot < info.y && g >= info.w && (ABLATE == 0 || g < n))
            {
                KeyT k;
                uint32_t v;
                s.buf.get((uint32_t) TILE + d * CSTRIDE + slot, k, v);
                dst_keys[g] = codec_out.decode(k);
                if (VALS) dst_vals[g] = v;
            }
        }
    }
    if (SEG) __syncthreads(); // the carry slots and the per-digit records are rewritten by the next sub-block
    } // sub-blocks
    seg_sh += nb;
    } while (SEG && seg_sh < seg_nshares); // shares
    if (STAMPS && lane == 0 && (wave == 0 || wave == WAVES - 1) && stamps)
    {
        // first and last wave of the workgroup: between them they show what a phase costs and what the barrier hides
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&stamps[(wave == 0 ? 0 : 8) + i], acc[i]);
    }
}

} // namespace glu_hip
