// radix_lds_bucket.hpp -- the in-LDS pass of a whole-key sort that ends in LDS (radix_lds_finish.hpp), round 6 form: ONE bucket round
// on unique words instead of two (six) ballot-ranked rounds, and whole 128-byte lines in and out.
//
// What the pass has to do (it replaces the reference's last 4 / 12 counting passes, glu/RadixSort.hpp:60-183, 289-333): order the
// pairs of one run -- equal top key bits, input order -- stably by the key bits [0, low_bits).  Round 5's kernel
// (radix_finish_sort_kernel) ranks 8 bits at a time with one ballot per digit bit: about 130 vector instructions per pair,
// which is what bounds it (profiles/r05/finish_pass_what_bounds_it.txt), and it loads and stores 4-byte elements at whatever
// alignment the run starts with (12 % / 27 % more bytes written than the pass owns).  This kernel:
//
//   * SLOTS.  Slot 0 of the tile is the 128-byte line boundary at or below the run's first pair; slot s is element
//     begin - front + s of the key array and of the value array (front = begin & 31: up to 31 slots in front of the run and
//     what is left of the last 16-byte piece behind it belong to the neighbouring runs -- loaded, never stored).  Every lane
//     loads 16 bytes at a time, a wave instruction covers 1 KiB of whole lines; stores likewise, element-wise only in the two
//     ragged pieces at the run's ends.
//   * VALUES never move with their keys: they wait in registers while the words are ordered, then go to vstage[slot] (16-byte LDS
//     stores, into the space the bucket table occupied) and are read once, by slot, at the end.
//   * WORDS.  What is sorted is one word per pair: (key bits [0, low_bits)) << SLOT_BITS | slot.  Words are UNIQUE, and their
//     order is the stable order of the pairs -- so any correct sort of the words gives the stable result, whatever order a
//     step hands equal digits in.  That is what lets the histogram be LDS atomics (their order across lanes is not specified
//     and does not matter here):
//       1. bucket = the word's top DIGIT_BITS bits (about as many buckets as the tile has slots: uniformly drawn keys leave 1 .. 1.5
//          words per bucket); a returning LDS atomic counts the bucket and hands the word a place in it;
//       2. one scan over the buckets; the words go to bucket start + place;
//       3. every bucket with more than one word is put in order by the thread that owns it: up to four (eight) words into
//          registers, a sorting network of min / max, back.  Exact, because words are unique.  (Round 6's first form let every
//          word count the smaller words of its bucket in a loop: a chain of dependent LDS reads as long as the longest bucket
//          any lane of the wave met -- slower than the ballots, profiles/r06/finish_bucket_first.txt.)
//     67 vector instructions per pair instead of 130 (SQ_INSTS_VALU, profiles/r06/finish_bucket_what_bounds_it.txt).
//   * CROWDED RUNS.  Keys whose low bits repeat (a run of a few distinct keys, keys that are multiples of 65536) fill few
//     buckets with many words, and step 3 is made for buckets of a few.  A word that finds kBucketMaxLen others in its bucket
//     proves the run crowded: the workgroup stops, stores nothing, and appends the run to a list (crowded_list_append,
//     radix_lds_finish.hpp) that radix_finish_sort_kernel -- round 5's ballot-ranked kernel, launched behind this one -- works off.
//     Keys of 25 .. 27 varying bits (9 .. 11 bits left to order, every key value a few times over in its run) are not crowded by
//     that measure, but their buckets hold two to four words each: the pass takes 1.1 ms for 2^28 pairs instead of 0.81.
//     (Both ways of ordering a crowded run inside this kernel were built and measured: the ballot rounds inlined cost the
//     uncrowded path 9 % in registers and scalar spills, profiles/r06/finish_bucket_variants.txt.)
//   * OUTPUT.  Lane j reads four consecutive ordered words (one 16-byte LDS load; the word positions are shifted by
//     front & 3 so that this is aligned on both sides), rebuilds the keys (run's upper bits | field), gathers the four values
//     by slot, and stores 16 bytes of keys and 16 of values, non-temporal.
//
// In place in the arrays that hold the data after the two top-bit passes (PassPlan::flip[pass]); the geometry comes from the
// PassPlan as before.  Runs longer than the tile are left to the segmented passes (long_ok) exactly as before.
#pragma once

#include "radix_lds_finish.hpp"

namespace glu_hip
{
constexpr uint32_t bucket_ceil_log2(uint32_t v)
{
    uint32_t b = 0;
    while ((1u << b) < v) b++;
    return b;
}
constexpr uint32_t bucket_floor_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

// a bucket longer than this makes the run a crowded one (step 3 sorts buckets in registers, networks for 4 and 8 words, and by
// insertion in the stage beyond: quadratic, one lane at work)
constexpr uint32_t kBucketMaxLen = 24;
// The value loads are issued behind the scatter of the words instead of with the key loads: 20 registers fewer while the words
// are ranked, and a crowded run never loads its values here.  4-byte keys 0.89 -> 0.81 ms at 2^28 pairs; 8-byte keys 1.41 -> 1.45
// (profiles/r06/finish_bucket_variants.txt), so by key size.  -DGLU_BUCKET_LATE_VALUES=0 / 1: tuning builds.
template<typename KeyT>
constexpr bool bucket_late_values()
{
#ifdef GLU_BUCKET_LATE_VALUES
    return GLU_BUCKET_LATE_VALUES != 0;
#else
    return sizeof(KeyT) == 4;
#endif
}

template<typename KeyT, int THREADS, int KPT, bool VALS>
struct BucketSmem
{
    using WordT = KeyT;
    static constexpr int WAVES = THREADS / kWave;
    static constexpr int TILE = THREADS * KPT;
    static constexpr int FRONT = 32;                              // slots in front of the run: what is left of a 128-byte line of 4-byte elements
    static constexpr int CAP = TILE + FRONT;                      // slots (a multiple of 4)
    static constexpr int SLOT_BITS = (int) bucket_ceil_log2(CAP); // 13 for 4640 slots, 14 for 9248
    static constexpr int NB = (int) bucket_floor_pow2(TILE);      // buckets: 4096 for a tile of 4608
    static constexpr int DIGIT_BITS = (int) bucket_ceil_log2(NB);
    static constexpr int BPT = NB / THREADS;                      // buckets per thread in the scan
    static constexpr int EPV = 16 / (int) sizeof(KeyT);           // keys per 16-byte piece
    static constexpr int KV = (CAP / EPV + THREADS - 1) / THREADS; // key pieces per lane
    static constexpr int VQ = (CAP / 4 + THREADS - 1) / THREADS;   // value pieces per lane
    static constexpr int ITEMS = KV * EPV;
    static_assert(CAP % 4 == 0 && BPT >= 4 && BPT % 4 == 0, "bucket geometry");
    union
    {
        alignas(16) uint32_t hist[NB + 4];           // bucket counts -> bucket start | length << 16 (steps 1 - 3)
        alignas(16) uint32_t vstage[VALS ? CAP : 4]; // then: the values by slot
    } a;
    alignas(16) WordT wstage[CAP + 12]; // the words in bucket order, then in order (positions shifted by front & 3)
    uint32_t scan_tmp[WAVES];
    uint32_t crowded; // a wave met a bucket longer than kBucketMaxLen
    KeyT run_hi; // the key bits from low_bits up: the same for every pair of the run
};
static_assert(sizeof(BucketSmem<uint32_t, 256, 18, true>) <= 40 * 1024, "four workgroups per CU");
static_assert(sizeof(BucketSmem<uint32_t, 512, 18, true>) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(BucketSmem<uint64_t, 512, 9, true>) <= 80 * 1024, "64-bit keys: two workgroups per CU");

// A workgroup per run (LOOP = false) or every gridDim.x-th run (LOOP = true: the geometries enqueued besides the expected one).
// XF: typed keys, encoded on load by the first top-bit pass, decoded on store here.
#ifndef GLU_BUCKET_WAVES_PER_SIMD
#define GLU_BUCKET_WAVES_PER_SIMD 4 // (tuning builds)
#endif
template<typename KeyT, int THREADS, int KPT, bool VALS, bool LOOP, bool XF = false>
__global__ __launch_bounds__(THREADS, GLU_BUCKET_WAVES_PER_SIMD) void radix_finish_bucket_kernel(KeyT* keys_a, uint32_t* vals_a, KeyT* keys_b, uint32_t* vals_b,
                                                                      const uint32_t* __restrict__ starts, uint32_t low_bits,
                                                                      const PassPlan* plan, uint32_t pass, uint32_t geometry,
                                                                      uint32_t key_xf = 0, uint32_t nruns = kFinishRuns,
                                                                      uint32_t* __restrict__ crowded = nullptr)
{
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    using Smem = BucketSmem<KeyT, THREADS, KPT, VALS>;
    using WordT = typename Smem::WordT;
    constexpr int WAVES = Smem::WAVES;
    constexpr int CAP = Smem::CAP, SB = Smem::SLOT_BITS, NB = Smem::NB, DB = Smem::DIGIT_BITS, BPT = Smem::BPT;
    constexpr int EPV = Smem::EPV, KV = Smem::KV, VQ = Smem::VQ, ITEMS = Smem::ITEMS;
    constexpr uint32_t SMASK = (1u << SB) - 1u;
    constexpr WordT INVALID = (WordT) ~(WordT) 0;
    constexpr bool kBucketLateValues = bucket_late_values<KeyT>();
    if (plan->top_bit) low_bits = plan->top_bit - 16u; // (the device chose the runs' bits: radix_sample_top_kernel)
    const KeyCodec<KeyT, XF> codec_out(key_xf);
    if (plan->finish != geometry) return; // (kernel-uniform: the device chose another geometry, or the ordinary passes)
    if (plan->finish_rounds) return;      // (kernel-uniform: few bits to order -- radix_finish_sort_kernel's, behind this kernel)

    KeyT* const keys = plan->flip[pass] ? keys_b : keys_a;
    uint32_t* const vals = plan->flip[pass] ? vals_b : vals_a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = starts[nruns];
    const KeyT low_mask = low_bits >= 8u * sizeof(KeyT) ? (KeyT) ~(KeyT) 0 : (KeyT) ((((KeyT) 1) << low_bits) - 1);
    // the bucket of a word: its top DIGIT_BITS bits (words have low_bits + SLOT_BITS bits; fewer than DIGIT_BITS: the word itself)
    const uint32_t dsh = low_bits + SB > (uint32_t) DB ? low_bits + SB - DB : 0u;

    for (uint32_t run = blockIdx.x; run < nruns; run += LOOP ? gridDim.x : nruns)
    {
    const uint32_t begin = starts[run], end = starts[run + 1];
    const uint32_t len = end - begin;
    if (len == 0 || (!XF && len == 1)) continue; // (workgroup-uniform; a single typed key still has to be decoded)
    if (len > (uint32_t) Smem::TILE) continue;   // (a run longer than the tile: the segmented passes')
    const uint32_t front = begin & 31u, f4 = front & 3u;
    const uint32_t abegin = begin - front;
    const uint32_t total = front + len; // slots [front, total) hold the run

    // ---- load: 16 bytes per lane and piece, keys and values into registers (the values wait there until the words are in order)
    u32x4_t kraw[KV];
#pragma unroll
    for (int r = 0; r < KV; r++)
    {
        const uint32_t q = r * THREADS + tid, e0 = abegin + q * EPV;
        kraw[r] = u32x4_t{0u, 0u, 0u, 0u};
        if (q * EPV < total && e0 + EPV <= n) kraw[r] = *reinterpret_cast<const u32x4_t*>(keys + e0);
    }
    u32x4_t vraw[VALS ? VQ : 1];
    auto load_values = [&]() {
#pragma unroll
        for (int r = 0; r < VQ; r++)
        {
            const uint32_t q = r * THREADS + tid, e0 = abegin + q * 4u;
            vraw[r] = u32x4_t{0u, 0u, 0u, 0u};
            if (q * 4u < total && e0 + 4u <= n) vraw[r] = *reinterpret_cast<const u32x4_t*>(vals + e0);
        }
        if (abegin + ((total + 3u) & ~3u) > n) // (workgroup-uniform: the array's last piece, see below)
        {
#pragma unroll
            for (int r = 0; r < VQ; r++)
            {
                const uint32_t q = r * THREADS + tid, e0 = abegin + q * 4u;
                if (q * 4u < total && e0 + 4u > n)
                {
                    uint32_t t[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) t[e] = e0 + e < n ? vals[e0 + e] : 0u;
                    __builtin_memcpy(&vraw[r], t, 16);
                }
            }
        }
    };
    if (VALS && !kBucketLateValues) load_values();
    // (the array's last piece when the count is not a multiple of the piece: only the array's last run can meet it -- its owner
    // loads it element by element)
    if (abegin + ((total + 3u) & ~3u) > n) // (workgroup-uniform)
    {
#pragma unroll
        for (int r = 0; r < KV; r++)
        {
            const uint32_t q = r * THREADS + tid, e0 = abegin + q * EPV;
            if (q * EPV < total && e0 + EPV > n)
            {
                KeyT t[EPV];
#pragma unroll
                for (int e = 0; e < EPV; e++) t[e] = e0 + e < n ? keys[e0 + e] : (KeyT) 0;
                __builtin_memcpy(&kraw[r], t, 16);
            }
        }
    }
    // (the bucket counts start at zero while the loads fly)
    {
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < BPT / 4; j++) *reinterpret_cast<u32x4_t*>(&s.a.hist[(j * THREADS + tid) * 4]) = z;
        if (tid == 0) *reinterpret_cast<u32x4_t*>(&s.a.hist[NB]) = z, s.crowded = 0u;
    }
    // ---- words.  live(r): piece r of this WAVE holds a slot of the run at all (wave-uniform, scalar: the fifth piece of a run of
    // 4100 pairs is a few lanes of wave 0 -- the other waves skip its code instead of running it with no lane enabled)
    const uint32_t wave_s = (uint32_t) __builtin_amdgcn_readfirstlane((int) wave);
    auto live = [&](int r) { return (r * THREADS + wave_s * kWave) * EPV < total; };
    WordT word[ITEMS];
#pragma unroll
    for (int r = 0; r < KV; r++)
    {
        KeyT t[EPV];
        __builtin_memcpy(t, &kraw[r], 16);
#pragma unroll
        for (int e = 0; e < EPV; e++)
        {
            const uint32_t slot = (r * THREADS + tid) * EPV + e;
            const bool valid = slot >= front && slot < total;
            word[r * EPV + e] = valid ? (WordT) (((WordT) (t[e] & low_mask) << SB) | (WordT) slot) : INVALID;
            if (slot == front) s.run_hi = (KeyT) (t[e] & ~low_mask);
        }
    }
    __syncthreads(); // the counts are zero
    // ---- 1. count: a returning LDS atomic per word (its place among the words of its bucket, in no particular order).  A word
    // whose place is kBucketMaxLen or more proves the run crowded: the wave says so in LDS and stops counting, the others look
    // before their second and third piece (same-address atomics serialise: a crowded run's count is the expensive part of it).
    uint32_t place[ITEMS];
    bool crowded_seen = false;
    {
        // (before any atomic: a wave whose first words crowd into one bucket -- four lanes or more with the bucket of the first
        // valid lane (eight in the small tiles), 0.02 expected for uniformly drawn keys -- says so at once: the atomics of a run of a few distinct keys
        // serialise 64-fold, 0.77 ms for 2^28 pairs of them where the loads alone take 0.3)
        const bool v0 = word[0] != INVALID;
        const uint64_t vm = __ballot(v0);
        if (vm != 0ull)
        {
            const uint32_t d0 = (uint32_t) (word[0] >> dsh);
            const uint32_t lead = (uint32_t) __builtin_amdgcn_readlane((int) d0, (int) __builtin_ctzll(vm));
            // (fewer buckets, more chance meetings: 64 words in 1024 buckets put four into one bucket 4 x 10^-5 of the time -- a few
            // runs of every 65536 -- and eight never)
            constexpr int kSameBucket = NB >= 4096 ? 4 : NB >= 2048 ? 6 : 8;
            if (__popcll(__ballot(v0 && d0 == lead)) >= kSameBucket)
            {
                crowded_seen = true;
                if (lane == 0) s.crowded = 1u;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < KV; r++)
    {
#pragma unroll
        for (int e = 0; e < EPV; e++) place[r * EPV + e] = 0u;
        if (live(r) && !crowded_seen)
        {
            if (r == 1 || r == 2) crowded_seen = __builtin_amdgcn_readfirstlane((int) *(volatile uint32_t*) &s.crowded) != 0;
            if (!crowded_seen)
            {
                uint32_t worst = 0;
#pragma unroll
                for (int e = 0; e < EPV; e++)
                {
                    const int i = r * EPV + e;
                    if (word[i] != INVALID) place[i] = atomicAdd(&s.a.hist[(uint32_t) (word[i] >> dsh)], 1u);
                    worst = max(worst, place[i]);
                }
                if (__ballot(worst >= kBucketMaxLen) != 0ull)
                {
                    crowded_seen = true;
                    if (lane == 0) s.crowded = 1u;
                }
            }
        }
    }
    __syncthreads();
    if (s.crowded != 0u) // (workgroup-uniform) radix_finish_sort_kernel's, launched behind this kernel; nothing has been stored
    {
        if (tid == 0) crowded_list_append(crowded, nruns, run);
        if (LOOP) __syncthreads();
        continue;
    }
    // ---- 2. scan of the bucket counts (thread t: buckets [t * BPT, (t + 1) * BPT)): bucket start | length << 16
    {
        uint32_t c[BPT];
#pragma unroll
        for (int j = 0; j < BPT / 4; j++)
        {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&s.a.hist[tid * BPT + j * 4]);
            c[j * 4 + 0] = v.x, c[j * 4 + 1] = v.y, c[j * 4 + 2] = v.z, c[j * 4 + 3] = v.w;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < BPT; j++)
        {
            const uint32_t t = c[j];
            c[j] = sum | (t << 16); // (start inside this thread's buckets | length: both below 2^16)
            sum += t;
        }
        uint32_t wtotal;
        const uint32_t excl = wave_exclusive_sum(sum, lane, wtotal);
        if (lane == 0) s.scan_tmp[wave] = wtotal;
        __syncthreads();
        uint32_t base = f4 + excl;
#pragma unroll
        for (int w = 0; w < WAVES; w++) base += (uint32_t) w < wave ? s.scan_tmp[w] : 0u;
#pragma unroll
        for (int j = 0; j < BPT / 4; j++)
        {
            const u32x4_t v = {base + c[j * 4 + 0], base + c[j * 4 + 1], base + c[j * 4 + 2], base + c[j * 4 + 3]};
            *reinterpret_cast<u32x4_t*>(&s.a.hist[tid * BPT + j * 4]) = v;
        }
    }
    __syncthreads();
    // the values go to their slots (16-byte LDS stores) once the bucket table, whose place they take, is done with
    auto values_to_slots = [&]() {
#pragma unroll
        for (int r = 0; r < VQ; r++)
        {
            const uint32_t q = r * THREADS + tid;
            if (q * 4u < total) *reinterpret_cast<u32x4_t*>(&s.a.vstage[q * 4u]) = vraw[r];
        }
    };
    {
        // ---- the words to bucket start + place
#pragma unroll
        for (int r = 0; r < KV; r++)
        {
            if (!live(r)) continue;
#pragma unroll
            for (int e = 0; e < EPV; e++)
            {
                const int i = r * EPV + e;
                if (word[i] != INVALID) s.wstage[(s.a.hist[(uint32_t) (word[i] >> dsh)] & 0xFFFFu) + place[i]] = word[i];
            }
        }
        __syncthreads();
        if (VALS && kBucketLateValues) load_values(); // (their latency passes under step 3)
        // ---- 3. buckets with more than one word, by the thread that owns them (bucket j * THREADS + tid: neighbouring lanes work on
        // neighbouring stretches of the stage): up to 4 (8) words into registers, pads ~0 behind them, a sorting network, back
        uint32_t hm_next = s.a.hist[tid];
#pragma unroll 1
        for (int j = 0; j < BPT; j++)
        {
            const uint32_t hm = hm_next;
            if (j + 1 < BPT) hm_next = s.a.hist[(j + 1) * THREADS + tid];
            const uint32_t st = hm & 0xFFFFu, m = hm >> 16;
            if (m >= 2u)
            {
                auto cx = [](WordT& a, WordT& b) {
                    const WordT lo = a < b ? a : b, hi = a < b ? b : a;
                    a = lo, b = hi;
                };
                if (m <= 4u)
                {
                    WordT x0 = s.wstage[st], x1 = s.wstage[st + 1], x2 = s.wstage[st + 2], x3 = s.wstage[st + 3];
                    x2 = m > 2u ? x2 : INVALID;
                    x3 = m > 3u ? x3 : INVALID;
                    cx(x0, x1), cx(x2, x3), cx(x0, x2), cx(x1, x3), cx(x1, x2);
                    s.wstage[st] = x0, s.wstage[st + 1] = x1;
                    if (m > 2u) s.wstage[st + 2] = x2;
                    if (m > 3u) s.wstage[st + 3] = x3;
                }
                else if (m <= 8u)
                {
                    WordT x[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) x[e] = s.wstage[st + e];
#pragma unroll
                    for (int e = 5; e < 8; e++) x[e] = m > (uint32_t) e ? x[e] : INVALID;
                    // (odd-even merge sort of 8: 19 exchanges)
                    cx(x[0], x[1]), cx(x[2], x[3]), cx(x[4], x[5]), cx(x[6], x[7]);
                    cx(x[0], x[2]), cx(x[1], x[3]), cx(x[4], x[6]), cx(x[5], x[7]);
                    cx(x[1], x[2]), cx(x[5], x[6]);
                    cx(x[0], x[4]), cx(x[1], x[5]), cx(x[2], x[6]), cx(x[3], x[7]);
                    cx(x[2], x[4]), cx(x[3], x[5]);
                    cx(x[1], x[2]), cx(x[3], x[4]), cx(x[5], x[6]);
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (m > (uint32_t) e) s.wstage[st + e] = x[e];
                }
                else
                {
                    // (9 .. kBucketMaxLen words: one bucket in 10^6 for uniformly drawn keys -- stable insertion in the stage)
                    for (uint32_t a = st + 1u; a < st + m; a++)
                    {
                        const WordT x = s.wstage[a];
                        uint32_t b = a;
                        while (b > st)
                        {
                            const WordT y = s.wstage[b - 1u];
                            if (y < x) break;
                            s.wstage[b] = y;
                            b--;
                        }
                        s.wstage[b] = x;
                    }
                }
            }
        }
        if (VALS)
        {
            __syncthreads(); // (the bucket table has been read)
            values_to_slots();
        }
    }
    __syncthreads();
    // ---- output: four consecutive positions per lane and piece, whole 16-byte pieces of keys and values
    {
        const KeyT run_hi = s.run_hi;
        const uint32_t pieces = (f4 + len + 3u) / 4u;
        const uint32_t ebase = begin - f4; // element of position 0 of the shifted stage (a multiple of 4)
        for (uint32_t j = tid; j < pieces; j += THREADS)
        {
            WordT w4[4];
            if constexpr (sizeof(WordT) == 4)
            {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&s.wstage[j * 4]);
                w4[0] = v.x, w4[1] = v.y, w4[2] = v.z, w4[3] = v.w;
            }
            else
            {
                const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(&s.wstage[j * 4]);
                const u32x4_t v1 = *reinterpret_cast<const u32x4_t*>(&s.wstage[j * 4 + 2]);
                w4[0] = (WordT) v0.x | ((WordT) v0.y << 32), w4[1] = (WordT) v0.z | ((WordT) v0.w << 32);
                w4[2] = (WordT) v1.x | ((WordT) v1.y << 32), w4[3] = (WordT) v1.z | ((WordT) v1.w << 32);
            }
            KeyT k4[4];
            uint32_t v4[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                k4[e] = codec_out.decode((KeyT) (run_hi | (KeyT) ((w4[e] >> SB) & low_mask)));
                v4[e] = VALS ? s.a.vstage[(uint32_t) w4[e] & SMASK] : 0u;
            }
            const uint32_t p0 = j * 4u; // shifted position of the piece's first element; the run's are [f4, f4 + len)
            if (p0 >= f4 && p0 + 4u <= f4 + len)
            {
                if constexpr (sizeof(KeyT) == 4)
                {
                    const u32x4_t kv = {(uint32_t) k4[0], (uint32_t) k4[1], (uint32_t) k4[2], (uint32_t) k4[3]};
                    __builtin_nontemporal_store(kv, reinterpret_cast<u32x4_t*>(keys + ebase + p0));
                }
                else
                {
                    const u32x4_t ka = {(uint32_t) k4[0], (uint32_t) ((uint64_t) k4[0] >> 32), (uint32_t) k4[1], (uint32_t) ((uint64_t) k4[1] >> 32)};
                    const u32x4_t kb = {(uint32_t) k4[2], (uint32_t) ((uint64_t) k4[2] >> 32), (uint32_t) k4[3], (uint32_t) ((uint64_t) k4[3] >> 32)};
                    __builtin_nontemporal_store(ka, reinterpret_cast<u32x4_t*>(keys + ebase + p0));
                    __builtin_nontemporal_store(kb, reinterpret_cast<u32x4_t*>(keys + ebase + p0 + 2));
                }
                if (VALS)
                {
                    const u32x4_t vv = {v4[0], v4[1], v4[2], v4[3]};
                    __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t*>(vals + ebase + p0));
                }
            }
            else
            {
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const uint32_t p = p0 + e;
                    if (p >= f4 && p < f4 + len)
                    {
                        __builtin_nontemporal_store(k4[e], &keys[ebase + p]);
                        if (VALS) __builtin_nontemporal_store(v4[e], &vals[ebase + p]);
                    }
                }
            }
        }
    }
    if (LOOP) __syncthreads(); // (the stages are reused)
    }
}

} // namespace glu_hip
