// glu_hip.hip -- RadixSort and the sharded sort of libglu_hip.so: the C ABI of include/glu_hip.h (gfx950 only, no CPU fallback).
// The library's other translation units: glu_core.hip, glu_sort_finish.hip, glu_scan_reduce.hip (glu_host.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <strings.h>
#include <unordered_map>
#include <vector>

#include "glu_host.hpp"
#include "glu_sort_launch.hpp"
#include "glu_sort_object.hpp"
#include "radix_sort_kernels.hpp"
#include "radix_scatter_lines.hpp"
#include "radix_pair_passes.hpp"
#include "radix_seg_passes.hpp"
#include "radix_lds_plan.hpp"

using namespace glu_hip;
using namespace glu_hip::host;

// ------------------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------------------
namespace
{

constexpr size_t kTuneMinKeyBytes = (size_t) 512 << 20; // scratch arrays from this size up are placed by measurement
glu_status tune_scratch_placement(glu_radix_sort_s* s, size_t count, size_t key_size);

// may_place: only the explicit prepare entry points (glu_radix_sort_prepare*, glu_dist_prepare) pass true.  The lazy call at
// the head of every sort, of a segmented sort and of glu_dist_sort_finish takes the plain allocation: a sort call, a
// collective or a stream capture never measures, never runs calibration sorts and never makes transient allocations.
glu_status sort_prepare(glu_radix_sort_s* s, size_t count, size_t key_size, bool with_vals = true, bool may_place = false)
{
    if (count <= 1) return GLU_OK;
    // large key + value scratch that has to be (re)allocated anyway: where the two arrays lie to each other decides between
    // discrete speeds of every pass, so a few placements are tried and the fastest is kept
    const bool grows = s->keys.size < count * key_size || (with_vals && s->vals.size < count * sizeof(uint32_t));
    if (grows && !s->tuning) // the arrays the last placement search chose (if any) are about to be replaced
        s->tuned_candidates = 0, s->tuned_ms = s->tuned_worst_ms = 0.0, s->tuned_spacer_mib = 0;
    if (may_place && with_vals && s->tune_scratch && !s->tuning && count * key_size >= kTuneMinKeyBytes && grows)
        GLU_TRY(tune_scratch_placement(s, count, key_size));
    GLU_TRY(s->keys.reserve(count * key_size));
    if (with_vals) GLU_TRY(s->vals.reserve(count * sizeof(uint32_t)));
    uint64_t cap = (uint64_t) g_dev.num_cus * kMaxBlocksPerCu;
    GLU_TRY(s->table.reserve((kMaxRadix * cap + kMaxRadix) * sizeof(uint32_t)));
    if (s->plan.size < sizeof(PassPlan))
    {
        GLU_TRY(s->plan.reserve(sizeof(PassPlan)));
        HIP_TRY(hipMemset(s->plan.ptr, 0, sizeof(PassPlan))); // never read uninitialised (glu_radix_sort_read_plan)
        HIP_TRY(hipDeviceSynchronize());                       // (the sort's stream does not wait for the null stream)
    }
    const size_t pair_from = std::min<size_t>(s->pair_min ? s->pair_min : kPairMinKeyBytes / key_size,
                                              s->lds_finish ? (s->finish_min ? s->finish_min : finish_min_count(key_size)) : (size_t) -1);
    if (count >= kPlanMinCount && count >= pair_from && s->pairs && !s->no_plan &&
        !s->no_lines && !s->force_small)
    {
        const size_t nb = (size_t) g_dev.num_cus;
        GLU_TRY(s->pair_t2.reserve((size_t) kPairRadix * nb * kPairRowWords * sizeof(uint32_t)));
        GLU_TRY(s->pair_table.reserve(((size_t) kPairRadix * nb + kPairRadix) * sizeof(uint32_t)));
        GLU_TRY(s->pair_ranges.reserve(nb * sizeof(uint2)));
        GLU_TRY(s->pair_sub.reserve((size_t) kPair4Radix * nb * kPairSub * sizeof(uint32_t)));
        if (s->lds_finish)
        {
            GLU_TRY(s->finish_lengths.reserve((size_t) kFinishRuns * sizeof(uint32_t)));
            GLU_TRY(s->finish_starts.reserve(((size_t) kFinishRuns + 1) * sizeof(uint32_t)));
            GLU_TRY(s->finish_crowded.reserve(crowded_list_words(kFinishRuns) * sizeof(uint32_t)));
            GLU_TRY(s->finish_outcomes.reserve(256 * sizeof(uint32_t)));
            GLU_TRY(s->pair_wide.reserve((size_t) nb * kPairWideStride * sizeof(uint32_t)));
            {
                // runs longer than the in-LDS pass's tile are sorted by segmented passes (radix_finish_long_runs_kernel)
                const uint32_t shares = (uint32_t) g_dev.num_cus * kLongRunsSharesPerWg;
                GLU_TRY(s->long_image.reserve((size_t) LongRunsLayout(shares).words * sizeof(uint32_t)));
                GLU_TRY(s->long_hdr.reserve(64));
                GLU_TRY(s->long_bits.reserve(2 * ((size_t) kLongRunsMax + shares) * sizeof(uint64_t)));
                GLU_TRY(s->table.reserve(((size_t) kLongRunsMax + shares) * 256 * sizeof(uint32_t)));
            }
            if (!s->finish_hint)
            {
                HIP_TRY(hipHostMalloc((void**) &s->finish_hint, 64));
                memset(s->finish_hint, 0, 64);
            }
        }
    }
    return GLU_OK;
}


// n <= one tile: the whole sort in a single workgroup / single launch (always 8-bit digits: the result does not
// depend on the digit width)
template<typename KeyT, int THREADS, int KPT, bool XF>
glu_status launch_single_block_xf(KeyT* keys, uint32_t* vals, size_t count, uint32_t first_bit, uint32_t end_bit,
                                  hipStream_t stream, uint32_t xform)
{
    using Smem = SingleBlockSmem<KeyT, 8, THREADS, KPT>;
    auto kern = radix_sort_single_block_kernel<KeyT, 8, THREADS, KPT, XF>;
    static std::once_flag lds_opt_in;
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);
    hipLaunchKernelGGL(kern, dim3(1), dim3(THREADS), sizeof(Smem), stream, keys, vals, (uint32_t) count, first_bit, end_bit, xform);
    HIP_TRY(hipGetLastError());
    return GLU_OK;
}

template<typename KeyT, int THREADS, int KPT>
glu_status launch_single_block(KeyT* keys, uint32_t* vals, size_t count, uint32_t first_bit, uint32_t end_bit,
                               hipStream_t stream, uint32_t xform)
{
    if (xform) return launch_single_block_xf<KeyT, THREADS, KPT, true>(keys, vals, count, first_bit, end_bit, stream, xform);
    return launch_single_block_xf<KeyT, THREADS, KPT, false>(keys, vals, count, first_bit, end_bit, stream, 0);
}

template<typename KeyT>
constexpr size_t single_block_limit() { return sizeof(KeyT) == 4 ? 1024 * 16 : 1024 * 8; } // 144 KB / 112 KB of LDS

template<typename KeyT>
glu_status sort_single_block(KeyT* keys, uint32_t* vals, size_t count, uint32_t first_bit, uint32_t end_bit,
                             hipStream_t stream, uint32_t xform)
{
    if (count <= 1024) return launch_single_block<KeyT, 256, 4>(keys, vals, count, first_bit, end_bit, stream, xform);
    // 1024 threads even for small tiles: the passes are latency chains inside one CU, 16 waves shorten every link
    // (2048 pairs: 14 -> 11 us, 4096: 20 -> 15 us, 8192: 30 -> 23 us against 256 threads x 16 / 1024 x 12)
    if (count <= 2048) return launch_single_block<KeyT, 1024, 2>(keys, vals, count, first_bit, end_bit, stream, xform);
    if (count <= 4096) return launch_single_block<KeyT, 1024, 4>(keys, vals, count, first_bit, end_bit, stream, xform);
    if (count <= 8192) return launch_single_block<KeyT, 1024, 8>(keys, vals, count, first_bit, end_bit, stream, xform);
    if constexpr (sizeof(KeyT) == 4)
    {
        if (count <= 1024 * 12) return launch_single_block<KeyT, 1024, 12>(keys, vals, count, first_bit, end_bit, stream, xform);
        return launch_single_block<KeyT, 1024, 16>(keys, vals, count, first_bit, end_bit, stream, xform);
    }
    else
        return launch_single_block<KeyT, 1024, 8>(keys, vals, count, first_bit, end_bit, stream, xform);
}


// The segmented passes over the LONG runs of a whole-key sort that ends in LDS (radix_finish_long_runs_kernel built their
// descriptors; hdr[0] = 0: there are none, or the sort was refused -- every kernel returns at once): key bits [0, 8) from the
// arrays that hold the data (PassPlan::flip[2], known on the device) into the other pair, bits [8, 16) back -- 4-byte keys.
// Round 6: 8-byte keys (six passes: up to 48 bits are left below the runs' bits), keys-only sorts, typed keys (the last pass
// decodes on store what the first top-bit pass encoded on load: the in-LDS pass, which does that for the other runs, leaves these alone).
template<typename KeyT, bool VALS, bool XF>
glu_status launch_long_run_passes(glu_radix_sort_s* s, KeyT* a_k, uint32_t* a_v, KeyT* b_k, uint32_t* b_v, size_t count, uint32_t key_xf,
                                  hipStream_t stream)
{
    using G = typename std::conditional<sizeof(KeyT) == 4 && VALS, SegLinesGeometry, LinesGeometry<KeyT, 8, VALS>>::type;
    constexpr int RADIX = 256;
    constexpr int RS = (G::KPT + 2) / 3;
    constexpr uint32_t kPasses = sizeof(KeyT) == 4 ? 2u : 6u; // (an even number: the runs come home)
    using Smem = LineSmem<KeyT, 8, G::THREADS, G::KPT, VALS>;
    auto scatter = radix_scatter_lines_kernel<KeyT, 8, G::THREADS, G::KPT, XF, VALS, 0, false, RS, true, true, 0, true>;
    static std::once_flag lds_opt_in;
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);
    // (kLongRunsSharesPerWg shares per workgroup, dealt round-robin: what is left of the long runs once those of one key value have
    // been emptied is spread over the chip)
    const uint32_t nwg = usable_cus(s), shares = nwg * kLongRunsSharesPerWg;
    const LongRunsLayout lay(shares);
    const uint32_t* image = (const uint32_t*) s->long_image.ptr;
    const uint32_t* hdr = (const uint32_t*) s->long_hdr.ptr;
    uint32_t* table = (uint32_t*) s->table.ptr;
    const PassPlan* plan = (const PassPlan*) s->plan.ptr;
    // (the first pass's count kernel also notes OR / AND of every sub-block's keys, its scan kernel empties the segments whose keys
    // agree on all the ordered bits: a long run of one key value -- every long run of a duplicate-heavy input -- stays where it is)
    KeyT* const sub_or = (KeyT*) s->long_bits.ptr;
    KeyT* const sub_and = sub_or + (kLongRunsMax + shares);
    for (uint32_t p = 0; p < kPasses; p++)
    {
        // (not typed keys: their long runs have to pass through the last pass, which decodes them)
        if (p == 0 && !XF)
        {
            hipLaunchKernelGGL((radix_seg_count_kernel<KeyT, 8, 1024, true>), dim3(nwg), dim3(1024), 0, stream, (const KeyT*) a_k, (const uint2*) image,
                               image + lay.off_first, table, 0u, 255u, hdr, 0u, kSegGateIfNot, (const KeyT*) b_k, plan, 2u, 0u, sub_or, sub_and, hdr + 3);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((radix_seg_scan_kernel<RADIX, KeyT>), dim3(512), dim3(RADIX), 0, stream, table, image + lay.off_list,
                               image + lay.off_start, hdr, 0u, kSegGateIfNot, hdr, (uint2*) const_cast<uint32_t*>(image), (const KeyT*) sub_or,
                               (const KeyT*) sub_and, plan, (uint32_t) (8 * sizeof(KeyT) - 16));
            HIP_TRY(hipGetLastError());
        }
        else
        {
            hipLaunchKernelGGL((radix_seg_count_kernel<KeyT, 8, 1024>), dim3(nwg), dim3(1024), 0, stream, (const KeyT*) a_k, (const uint2*) image,
                               image + lay.off_first, table, p * 8u, 255u, hdr, 0u, kSegGateIfNot, (const KeyT*) b_k, plan, 2u, p, (KeyT*) nullptr, (KeyT*) nullptr, hdr + 3);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((radix_seg_scan_kernel<RADIX>), dim3(512), dim3(RADIX), 0, stream, table, image + lay.off_list,
                               image + lay.off_start, hdr, 0u, kSegGateIfNot, hdr); // (workgroups loop over the device-counted segments)
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(scatter, dim3(nwg), dim3(G::THREADS), sizeof(Smem), stream, (const KeyT*) a_k, (const uint32_t*) a_v, b_k, b_v,
                           (const uint32_t*) table, (const uint32_t*) nullptr, (uint32_t) count, p * 8u, 255u, 0u, (unsigned long long*) nullptr,
                           XF && p + 1 == kPasses ? key_xf << 2 : 0u, const_cast<PassPlan*>(plan), 2u, (const uint2*) image, p,
                           image + lay.off_first, hdr, 0u, kSegGateIfNot, hdr + 3);
        HIP_TRY(hipGetLastError());
    }
    return GLU_OK;
}
template<typename KeyT>
glu_status launch_long_run_passes_any(glu_radix_sort_s* s, KeyT* a_k, uint32_t* a_v, KeyT* b_k, uint32_t* b_v, size_t count, uint32_t key_xf,
                                      hipStream_t stream)
{
    if (a_v) return key_xf != KEY_XF_NONE ? launch_long_run_passes<KeyT, true, true>(s, a_k, a_v, b_k, b_v, count, key_xf, stream)
                                          : launch_long_run_passes<KeyT, true, false>(s, a_k, a_v, b_k, b_v, count, key_xf, stream);
    return key_xf != KEY_XF_NONE ? launch_long_run_passes<KeyT, false, true>(s, a_k, nullptr, b_k, nullptr, count, key_xf, stream)
                                 : launch_long_run_passes<KeyT, false, false>(s, a_k, nullptr, b_k, nullptr, count, key_xf, stream);
}

// vals == nullptr: keys-only sort (no value traffic, no value scratch)
template<typename KeyT>
glu_status sort_bits(glu_radix_sort_s* s, KeyT* keys, uint32_t* vals, size_t count, uint32_t first_bit, uint32_t end_bit,
                     hipStream_t stream, uint32_t key_xf = KEY_XF_NONE)
{
    if (count <= 1 || first_bit >= end_bit) return GLU_OK; // RadixSort.hpp:278-279
    if (count > 0xFFFF0000ull) return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu does not fit 32-bit indexing", count);
    if (((uintptr_t) keys % sizeof(KeyT)) != 0 || (vals && ((uintptr_t) vals % sizeof(uint32_t)) != 0))
        return fail(GLU_ERROR_INVALID_ARGUMENT, "key/value arrays must be aligned to their element size");
    GLU_TRY(sort_prepare(s, count, sizeof(KeyT), vals != nullptr)); // RadixSort.hpp:281 (no-op when prepared)
    s->last_planned = false; // (what glu_radix_sort_read_plan / read_finish report is about THIS sort)
    s->last_finish_attempted = false;
    s->last_finish_long_ok = false;

    if (count <= single_block_limit<KeyT>() && !s->no_single_block)
    {
        s->mark(stream); // profiling: booked as one "scatter" launch (count / scan intervals are empty)
        s->mark(stream);
        s->mark(stream, true);
        GLU_TRY(sort_single_block<KeyT>(keys, vals, count, first_bit, end_bit, stream, key_xf | (key_xf << 2)));
        s->mark(stream, true);
        return GLU_OK;
    }

    KeyT* kbuf[2] = {keys, (KeyT*) s->keys.ptr};
    uint32_t* vbuf[2] = {vals, vals ? (uint32_t*) s->vals.ptr : nullptr};
    int cur = 0;
    s->cur_kind = 0;
    s->cur_behind = false; // (a call that failed half-way must not leave the next one's profile marks switched off)
    s->cur_seq = 0;
    // large sorts: device-side pass plan (constant-digit passes are skipped, the arrays' roles follow on the device)
    const bool planned = count >= kPlanMinCount && !s->no_plan;
    s->last_planned = planned;
    // the passes: digit positions, key transforms (typed keys: encode on the first pass's loads, decode on the last
    // pass's stores), and which passes share one count kernel (radix_pair_passes.hpp)
    struct PassDesc
    {
        uint32_t shift, bits, xform;
        int pair_role;
    };
    PassDesc passes[kPlanMaxPasses];
    uint32_t num_passes = 0;
    for (uint32_t shift = first_bit; shift < end_bit;)
    {
        uint32_t bits = std::min<uint32_t>(s->digit_bits, end_bit - shift);
        if (sizeof(KeyT) == 8 && shift < 32 && shift + bits > 32) bits = 32 - shift; // a digit stays inside one key word
        const uint32_t xform = (shift == first_bit ? key_xf : 0u) | (shift + bits >= end_bit ? key_xf << 2 : 0u);
        if (num_passes == kPlanMaxPasses) return fail(GLU_ERROR_INVALID_ARGUMENT, "too many passes");
        passes[num_passes++] = PassDesc{shift, bits, xform, 0};
        shift += bits;
    }
    // A leader is a pass of the line kernel that does not encode keys on load; its follower is the pass after it, of the
    // line kernel of the same digit width (wider than 4 bits: the 8-bit kernels, else the 4-bit ones).
    const bool pair_tables = planned && s->pairs && s->pair_t2.ptr;
    const bool pairs_ok = pair_tables && count >= (s->pair_min ? s->pair_min : kPairMinKeyBytes / sizeof(KeyT));
    // Whole keys, 8-bit digits, line passes, from finish_min_count elements: the sort first tries to end in LDS
    // (radix_lds_finish.hpp) -- the two top-bit passes and the in-LDS pass are enqueued in front of the ordinary passes, and the
    // device runs one of the two sequences.
    uint32_t finish_kpt = 0; // (the geometry that suits uniform keys; 0: no attempt)
    uint32_t finish_top_bit = end_bit;
    if (pair_tables && s->lds_finish && s->finish_starts.ptr && first_bit == 0 &&
        end_bit == 8 * sizeof(KeyT) && s->digit_bits == 8 && num_passes == sizeof(KeyT) && count >= (s->finish_min ? s->finish_min : finish_min_count(sizeof(KeyT))) &&
        lines_applicable<KeyT, 8>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count))
        finish_kpt = finish_geometry_for(count);
    // The runs' key bits are chosen on the device from a sample of the keys (untyped keys: radix_sample_top_kernel); the host only
    // remembers, as a floor for the next sample, the exact top bit of an attempt that was refused because its sample had missed
    // a varying bit.  GLU_HIP_SORT_DEVICE_TOP=0: the host guesses from the object's last attempt (round 4, below).
    const bool device_top = finish_kpt && s->device_top && key_xf == KEY_XF_NONE && !s->no_bit_shortcut;
    if (finish_kpt && s->finish_hint && device_top)
    {
        const uint32_t seen = s->finish_seq ? __atomic_load_n(s->finish_hint, __ATOMIC_ACQUIRE) : 0u;
        if (s->finish_wait == 0 && s->finish_seq && (seen >> 3) == s->finish_seq) // the last attempt's outcome has arrived
        {
            if (seen & 7u) s->finish_last_geo = seen & 7u; // the tile it took
            s->finish_last_refused = (seen & 7u) == 0u;
            if (s->finish_seq != s->finish_seq_acted_on)    // (acted on once)
            {
                s->finish_seq_acted_on = s->finish_seq;
                bool range_miss = false;
                if (__atomic_load_n(s->finish_hint + 3, __ATOMIC_ACQUIRE) == s->finish_seq)
                {
                    const uint64_t varying = (uint64_t) s->finish_hint[1] | ((uint64_t) s->finish_hint[2] << 32);
                    const uint32_t exact = varying ? 64u - (uint32_t) __builtin_clzll(varying) : 0u, used = s->finish_hint[4];
                    range_miss = !(seen & 7u) && exact > used;
                    if (range_miss)
                        s->top_floor = exact; // the sample missed a bit that varies: the next one starts from here
                    else if (exact < s->top_floor)
                        s->top_floor = 0;     // (other keys now)
                }
                if (!(seen & 7u) && !range_miss && s->finish_backoff) s->finish_wait = s->finish_backoff; // refused for its runs: do not ask again for a while
            }
        }
        if (s->finish_wait)
        {
            s->finish_wait--;
            finish_kpt = 0;
        }
    }
    else if (finish_kpt && s->finish_hint)
    {
        const uint32_t seen = s->finish_seq ? __atomic_load_n(s->finish_hint, __ATOMIC_ACQUIRE) : 0u;
        if (s->finish_wait == 0 && s->finish_seq && (seen >> 3) == s->finish_seq) // the last attempt's outcome has arrived
        {
            if (seen & 7u) s->finish_last_geo = seen & 7u; // the tile it took
            s->finish_last_refused = (seen & 7u) == 0u;
            if (s->finish_seq != s->finish_seq_acted_on)    // (acted on once)
            {
                s->finish_seq_acted_on = s->finish_seq;
                const uint32_t was = s->finish_top ? s->finish_top : end_bit;
                uint32_t top = was;
                if (__atomic_load_n(s->finish_hint + 3, __ATOMIC_ACQUIRE) == s->finish_seq)
                {
                    // which key bits varied: the runs of the next attempt are the values of the top 16 of those
                    const uint64_t varying = (uint64_t) s->finish_hint[1] | ((uint64_t) s->finish_hint[2] << 32);
                    top = varying ? 64u - (uint32_t) __builtin_clzll(varying) : 16u;
                    top = std::min<uint32_t>(std::max<uint32_t>(top, 16u), end_bit);
                    if (sizeof(KeyT) == 8 && top > 32 && top < 48 && top != 40) top = top < 40 ? 40 : 48; // a digit stays inside one key word
                }
                if (top != was)
                    s->finish_top = top; // the assumption changes: what the last attempt was told says nothing about the next
                else if (!(seen & 7u) && s->finish_backoff)
                    s->finish_wait = s->finish_backoff; // refused under the same assumption: do not ask again for a while
            }
        }
        if (s->finish_wait)
        {
            s->finish_wait--;
            finish_kpt = 0;
        }
    }
    if (finish_kpt)
    {
        s->finish_seq = s->finish_seq >= 0x0FFFFFFFu ? 1u : s->finish_seq + 1;
        s->cur_seq = s->finish_seq;
        for (uint32_t i = num_passes; i-- > 0;) passes[i + 2] = passes[i];
        // (typed keys and sorts that do not collect which bits vary: the whole key's top bits)
        finish_top_bit = !device_top && key_xf == KEY_XF_NONE && !s->no_bit_shortcut && s->finish_top ? std::min<uint32_t>(s->finish_top, end_bit) : end_bit;
        passes[0] = PassDesc{finish_top_bit - 16u, 8u, key_xf, 0}; // (typed keys: encoded on load here, decoded on store by the in-LDS pass)
        passes[1] = PassDesc{finish_top_bit - 8u, 8u, 0u, 0};
        num_passes += 2;
    }
    // runs longer than the tile: segmented passes over just them (4-byte untyped keys with values, all 16 low bits inside two digits)
    // (every key type since round 6; the passes cover 16 low bits of 4-byte keys, 48 of 8-byte keys)
    const bool finish_long_ok = finish_kpt && s->long_runs && s->long_image.ptr && finish_top_bit >= 16 &&
                                finish_top_bit - 16u <= (sizeof(KeyT) == 4 ? 16u : 48u);
    s->last_finish_long_ok = finish_long_ok;
    s->last_device_top = finish_kpt && device_top;
    s->last_finish_top = finish_kpt ? finish_top_bit : 0u;
    s->last_finish_attempted = finish_kpt != 0;
    const uint32_t finish_last = std::min<uint32_t>(finish_kpt + 2, kFinishGeometries); // the larger tiles enqueued behind it
    // the one that gets a workgroup per run: what this object's last sort took if that is among them, else the uniform-keys one
    const uint32_t finish_expected = s->finish_last_geo >= finish_kpt && s->finish_last_geo <= finish_last ? s->finish_last_geo : finish_kpt;
    s->last_finish_capacity = finish_kpt ? finish_geometry_capacity(finish_last) : 0u;
    if (pairs_ok || finish_kpt)
    {
        const bool lines8 = lines_applicable<KeyT, 8>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count);
        const bool lines4 = lines_applicable<KeyT, 4>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count);
        for (uint32_t i = 0; i + 1 < num_passes;)
        {
            const bool wide = passes[i].bits > 4;
            // (below the size from which ordinary passes pair, only the two top-bit passes of an attempt to end in LDS do)
            // (a leader does not encode keys on load -- except the first top-bit pass of an attempt, whose 8-bit pair-count
            // kernel has the encoding instantiation)
            if ((pairs_ok || (finish_kpt && i == 0)) && wide == (passes[i + 1].bits > 4) && (wide ? lines8 : lines4) &&
                ((passes[i].xform & 3u) == 0 || (finish_kpt && i == 0)))
            {
                passes[i].pair_role = 1;
                passes[i + 1].pair_role = 2;
                i += 2;
            }
            else
                i += 1;
        }
    }
    if (planned)
    {
        // per sort: no follower counts for itself yet, nothing is known about the key bits
        PassPlan* plan = (PassPlan*) s->plan.ptr;
        static_assert(offsetof(PassPlan, sample_done) + sizeof(plan->sample_done) <= sizeof(PassPlan) && offsetof(PassPlan, sample_done) + 64 > sizeof(PassPlan),
                      "sample_done is the last member of the zeroed tail of the plan");
        static_assert(offsetof(PassPlan, pair_fallback) % 64 == 0 && sizeof(PassPlan) % 64 == 0, "one aligned fill");
        HIP_TRY(hipMemsetAsync(plan->pair_fallback, 0, sizeof(PassPlan) - offsetof(PassPlan, pair_fallback), stream));
        if (finish_kpt && device_top)
        {
            hipLaunchKernelGGL((radix_sample_top_kernel<KeyT>), dim3(kSampleTopBlocks), dim3(256), 0, stream, (const KeyT*) kbuf[0], (uint32_t) count, end_bit,
                               std::min<uint32_t>(s->top_floor, end_bit), plan);
            HIP_TRY(hipGetLastError());
        }
    }
    if (planned)
        for (uint32_t i = 0; i < (uint32_t) kPlanMaxPasses; i++) s->last_pair_roles[i] = i < num_passes ? (uint32_t) passes[i].pair_role : 0u;
    // (see glu_radix_sort_s::side)
    // Four cross-stream dependencies cost 0.1 ms that only long kernels hide: a sort whose attempt is refused and whose
    // ordinary passes skip (2^28 all-zero keys, the reference's benchmark input: 0.22 ms on one queue, 0.35 forked) is
    // better off on one queue -- so an object whose last known attempt was refused does not fork (the first sort of an object does).
    const bool fork = planned && finish_kpt && s->fork_behind && !s->finish_last_refused && s->ensure_side();
    uint32_t pass = 0;
    for (; pass < num_passes; pass++)
    {
        const uint32_t shift = passes[pass].shift, bits = passes[pass].bits, xform = passes[pass].xform;
        if (planned)
        {
            PlanArgs pa;
            pa.plan = (PassPlan*) s->plan.ptr;
            pa.pass = pass;
            pa.may_skip = xform == 0; // encode / decode passes run whatever the data looks like
            // the first pass's count kernel notes which key bits vary at all (untyped keys: what is sorted is the raw
            // bit pattern); a later pass whose digit cannot vary then knows that it is an identity before it counts
            if (key_xf == KEY_XF_NONE && !s->no_bit_shortcut) pa.flags = pass == 0 ? kPlanCollectBits : (pa.may_skip ? kPlanShortcut : 0u);
            pa.pair_role = passes[pass].pair_role;
            if (pa.pair_role == 1)
            {
                pa.shift2 = passes[pass + 1].shift;
                pa.bits2 = passes[pass + 1].bits;
            }
            s->cur_kind = finish_kpt && pass < 2 ? 1 : 0;
            pa.behind_attempt = finish_kpt && pass >= 2;
            s->cur_behind = pa.behind_attempt; // (light profiling: no events around the sequence that is expected to return at once)
            if (finish_kpt && pass == 0)
            {
                pa.finish_geo_first = finish_kpt;
                pa.finish_geo_last = finish_last;
                pa.finish_first_ordinary = 2;
                pa.finish_num_ordinary = num_passes - 2;
                pa.finish_seq = s->finish_seq;
                pa.finish_top_bit = finish_top_bit;
                pa.finish_key_bits = end_bit;
                pa.finish_long_ok = finish_long_ok;
                pa.fork_side = fork;
            }
            pa.unitsum_side = fork && pass == 1 && pa.pair_role == 2 && bits == 8;
            GLU_TRY(dispatch_pass<KeyT>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count, shift, bits, nullptr,
                                        fork && pa.behind_attempt ? s->side : stream, xform, pa));
            {
                if (finish_kpt && pass == 1)
                {
                    if (fork) HIP_TRY(hipEventRecord(s->ev_fork2, stream)); // (the data is where the in-LDS pass and the long-run passes find it)
                    // the in-LDS pass, in place on whichever pair of arrays holds the data now (returns at once if the
                    // device chose the ordinary passes, which follow)
                    s->cur_kind = 2;
                    s->cur_behind = false;
                    s->mark(stream);
                    s->mark(stream);
                    const FinishMarks finish_marks{[](void* object, hipStream_t st) { ((glu_radix_sort_s*) object)->mark(st, true); }, s};
#define GLU_LAUNCH_FINISH(VALS_, XF_)                                                                                             \
    GLU_TRY((launch_finish<KeyT, VALS_, XF_>(kbuf[0], VALS_ ? vbuf[0] : nullptr, kbuf[1], VALS_ ? vbuf[1] : nullptr,              \
                                             (const uint32_t*) s->finish_starts.ptr, finish_kpt, finish_last, finish_expected,    \
                                             finish_top_bit - 16u, pa.plan, 2u, key_xf, stream, s->finish_rank_bits,              \
                                             (uint32_t*) s->finish_crowded.ptr, finish_marks)))
                    if (vals)
                    {
                        if (key_xf != KEY_XF_NONE) GLU_LAUNCH_FINISH(true, true);
                        else GLU_LAUNCH_FINISH(true, false);
                    }
                    else
                    {
                        if (key_xf != KEY_XF_NONE) GLU_LAUNCH_FINISH(false, true);
                        else GLU_LAUNCH_FINISH(false, false);
                    }
#undef GLU_LAUNCH_FINISH
                    if (finish_long_ok && !fork)
                        GLU_TRY(launch_long_run_passes_any<KeyT>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count, key_xf, stream));
                }
            }
        }
        else
        {
            GLU_TRY(dispatch_pass<KeyT>(s, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1], count, shift, bits, nullptr,
                                        stream, xform));
            cur ^= 1;
        }
    }
    s->cur_behind = false;
    if (fork)
    {
        // the side stream: (behind the ordinary passes) the segmented passes over long runs, beside the in-LDS pass; then it joins
        HIP_TRY(hipStreamWaitEvent(s->side, s->ev_fork2, 0));
        if (finish_long_ok) GLU_TRY(launch_long_run_passes_any<KeyT>(s, kbuf[0], vbuf[0], kbuf[1], vbuf[1], count, key_xf, s->side));
        HIP_TRY(hipEventRecord(s->ev_join, s->side));
        HIP_TRY(hipStreamWaitEvent(stream, s->ev_join, 0));
    }
    if (planned)
    {
        // the data is home unless an odd number of passes ran: decided and, if need be, copied on the device
        const dim3 grid((uint32_t) std::min<size_t>((count + 255) / 256, (size_t) g_dev.num_cus * 16));
        if (vals)
            hipLaunchKernelGGL((radix_finalize_kernel<KeyT, true>), grid, dim3(256), 0, stream, kbuf[0], vbuf[0], (const KeyT*) kbuf[1],
                               (const uint32_t*) vbuf[1], (uint32_t) count, (const PassPlan*) s->plan.ptr, pass, finish_kpt ? 2u : 0u);
        else
            hipLaunchKernelGGL((radix_finalize_kernel<KeyT, false>), grid, dim3(256), 0, stream, kbuf[0], (uint32_t*) nullptr,
                               (const KeyT*) kbuf[1], (const uint32_t*) nullptr, (uint32_t) count, (const PassPlan*) s->plan.ptr, pass,
                               finish_kpt ? 2u : 0u);
        HIP_TRY(hipGetLastError());
        return GLU_OK;
    }
    if (cur == 1)
    {
        // odd number of passes: the reference would leave the result in its private scratch (documented
        // deviation in glu_hip.h) -- bring it home
        HIP_TRY(hipMemcpyAsync(keys, kbuf[1], count * sizeof(KeyT), hipMemcpyDeviceToDevice, stream));
        if (vals) HIP_TRY(hipMemcpyAsync(vals, vbuf[1], count * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    }
    return GLU_OK;
}

// the reference's interface: num_steps 4-bit steps from bit 0 (0 or more than the key holds = the whole key,
// RadixSort.hpp:289,332)
template<typename KeyT>
glu_status sort_run(glu_radix_sort_s* s, KeyT* keys, uint32_t* vals, size_t count, size_t num_steps, hipStream_t stream,
                    uint32_t key_xf = KEY_XF_NONE)
{
    constexpr size_t kMaxSteps = sizeof(KeyT) * 2; // 4-bit steps: 8 for 32-bit keys
    const size_t steps = (num_steps == 0 || num_steps > kMaxSteps) ? kMaxSteps : num_steps;
    return sort_bits<KeyT>(s, keys, vals, count, 0u, (uint32_t) steps * 4, stream, key_xf);
}

// Placement of the two large scratch arrays by measurement (sort_prepare).  Two 1 GiB arrays that hipMalloc hands out one
// after the other lie 1 GiB + 2 MiB apart, and whether the key scratch and the value scratch then collide in the memory
// system is a property of where they fell, the same for every caller array the sort is later given (tools/placement_scratch.py,
// profiles/r03/placement_scratch_spacers.txt: of eight sorter objects in one process the same three sort every caller pair
// in 3.76-3.80 ms and the others in 3.9-4.05 ms).  User space cannot choose physical pages, but it can choose among
// allocations: the value array is allocated behind spacers of 0 .. 6 GiB, every candidate sorts a scratch copy of
// pseudo-random pairs twice (events on the library queue), the fastest pair of arrays is kept and everything else freed.
// One-off, and only inside the explicit prepare entry points (glu_radix_sort_prepare*, glu_dist_prepare; the reference's
// prepare_internal_buffers is where its allocations happen too).  A sort on an object that was not prepared for its size
// grows the scratch with two plain hipMallocs and reports zero candidates.  GLU_HIP_SCRATCH_TUNE=0 switches it off.
__global__ void tune_fill_kernel(uint32_t* __restrict__ p, size_t words, uint32_t salt)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x)
    {
        uint32_t x = (uint32_t) i * 2654435761u + salt;
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
        p[i] = x;
    }
}

glu_status tune_scratch_placement(glu_radix_sort_s* s, size_t count, size_t key_size)
{
    // (grow-only: an array that is already larger than this count needs keeps its size)
    const size_t kbytes = std::max(count * key_size, s->keys.size), vbytes = std::max(count * sizeof(uint32_t), s->vals.size);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return GLU_OK;
    // up to 16 candidates, the value array behind spacers of 0, 0.5 .. 7.5 GiB (which spacer wins differs from process to
    // process: 24 candidates on two devices showed no period, about one in four is fast).  The search ends as soon as one
    // candidate of at least eight is 7 % faster than the slowest seen -- the speeds are discrete, fast and slow lie 5-8 %
    // apart, and the fast ones differ among themselves by 2-3 %; eight candidates take 0.13 s -- or after a second: on some devices no placement out of 32 is
    // fast, and spacers of many GiB take a quarter of a second each to allocate and free (7.9 s for 32 candidates up to
    // 15.5 GiB).  GLU_HIP_SCRATCH_TUNE_LIST=step_mib:count[:first_mib] fixes the list (no early end).
    size_t step_mib = 512, candidates = 16, first_mib = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const char* const tune_list = glu_env("GLU_HIP_SCRATCH_TUNE_LIST");
    const bool fixed_list = tune_list != nullptr;
    if (const char* e = tune_list) sscanf(e, "%zu:%zu:%zu", &step_mib, &candidates, &first_mib);
    std::vector<size_t> spacers_mib;
    for (size_t i = 0; i < std::max<size_t>(candidates, 1); i++) spacers_mib.push_back(first_mib + i * step_mib);
    // room for the caller-side copies, one candidate with its spacer and the best so far (twice over, to be safe)
    const size_t need_b = 2 * (2 * (kbytes + vbytes) + (spacers_mib.back() << 20)) + ((size_t) 1 << 30);
    s->tuned_candidates = 0, s->tuned_ms = s->tuned_worst_ms = 0.0, s->tuned_spacer_mib = 0;
    if (free_b < need_b)
    {
        if (glu_verbose())
            fprintf(stderr, "[glu_hip] scratch placement skipped: %zu MiB free, the search wants %zu MiB; plain allocation\n", free_b >> 20,
                    need_b >> 20);
        return GLU_OK;
    }
    // the calibration sorts run on the library queue and rewrite this object's tables: nothing of the caller's may still be
    // using them on another stream
    if (hipDeviceSynchronize() != hipSuccess) return fail(GLU_ERROR_DEVICE, "hipDeviceSynchronize failed before the scratch placement");
    hipStream_t st = g_dev.queue;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Cand { void* k = nullptr; void* v = nullptr; double ms = 1e30; uint32_t spacer = 0; } best;
    double worst = 0;
    uint32_t tried = 0;
    s->tuning = true;
    const int was_profiling = s->profiling;
    s->profiling = 0; // the calibration sorts are not the caller's
    s->keys.release();
    s->vals.release();
    auto cleanup = [&](glu_status status) {
        s->profiling = was_profiling;
        (void) hipStreamSynchronize(st);
        if (a) (void) hipFree(a);
        if (b) (void) hipFree(b);
        if (e0) (void) hipEventDestroy(e0);
        if (e1) (void) hipEventDestroy(e1);
        s->keys.ptr = best.k, s->keys.size = best.k ? kbytes : 0;
        s->vals.ptr = best.v, s->vals.size = best.v ? vbytes : 0;
        s->tuning = false;
        s->last_planned = false; // glu_radix_sort_read_plan: the calibration sorts were not the caller's
        return status;
    };
#define TUNE_TRY(expr)                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return cleanup(fail(GLU_ERROR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)));   \
    } while (0)
    TUNE_TRY(hipMalloc(&a, kbytes));
    TUNE_TRY(hipMalloc(&b, vbytes));
    TUNE_TRY(hipEventCreate(&e0));
    TUNE_TRY(hipEventCreate(&e1));
    for (size_t spacer_mib : spacers_mib)
    {
        void *k = nullptr, *sp = nullptr, *v = nullptr;
        if (hipMalloc(&k, kbytes) != hipSuccess)
        {
            (void) hipGetLastError();
            break; // out of memory: the best so far (or the plain allocation of sort_prepare, which reports the failure)
        }
        if (spacer_mib && hipMalloc(&sp, spacer_mib << 20) != hipSuccess)
        {
            (void) hipGetLastError();
            (void) hipFree(k);
            continue;
        }
        if (hipMalloc(&v, vbytes) != hipSuccess)
        {
            (void) hipGetLastError();
            (void) hipFree(k);
            if (sp) (void) hipFree(sp);
            continue;
        }
        if (sp) (void) hipFree(sp); // the arrays stay where they are
        s->keys.ptr = k, s->keys.size = kbytes;
        s->vals.ptr = v, s->vals.size = vbytes;
        double ms = 1e30;
        glu_status status = GLU_OK;
        for (int rep = 0; rep < 3 && status == GLU_OK; rep++) // the first run is a warm-up
        {
            hipLaunchKernelGGL(tune_fill_kernel, dim3(g_dev.num_cus * 8), dim3(256), 0, st, (uint32_t*) a, kbytes / 4, 0x5EEDu + rep);
            hipLaunchKernelGGL(tune_fill_kernel, dim3(g_dev.num_cus * 8), dim3(256), 0, st, (uint32_t*) b, vbytes / 4, 77u);
            (void) hipEventRecord(e0, st);
            status = key_size == 8 ? sort_run<uint64_t>(s, (uint64_t*) a, (uint32_t*) b, count, 0, st)
                                   : sort_run<uint32_t>(s, (uint32_t*) a, (uint32_t*) b, count, 0, st);
            (void) hipEventRecord(e1, st);
            if (status == GLU_OK && hipEventSynchronize(e1) == hipSuccess && rep > 0)
            {
                float t = 0;
                if (hipEventElapsedTime(&t, e0, e1) == hipSuccess) ms = std::min(ms, (double) t);
            }
        }
        s->keys.ptr = nullptr, s->keys.size = 0;
        s->vals.ptr = nullptr, s->vals.size = 0;
        if (status != GLU_OK)
        {
            (void) hipStreamSynchronize(st);
            (void) hipFree(k);
            (void) hipFree(v);
            return cleanup(status);
        }
        if (ms < 1e29) worst = std::max(worst, ms);
        if (glu_verbose()) fprintf(stderr, "[glu_hip]   candidate: keys %p values %p (spacer %zu MiB): %.3f ms\n", k, v, spacer_mib, ms);
        if (ms < best.ms)
        {
            if (best.k) (void) hipFree(best.k);
            if (best.v) (void) hipFree(best.v);
            best.k = k, best.v = v, best.ms = ms, best.spacer = (uint32_t) spacer_mib;
        }
        else
        {
            (void) hipFree(k);
            (void) hipFree(v);
        }
        tried++;
        if (!fixed_list && tried >= 8 && best.ms <= 0.93 * worst) break;
        if (!fixed_list && std::chrono::steady_clock::now() - t_begin > std::chrono::milliseconds(1000)) break;
    }
#undef TUNE_TRY
    s->tuned_spacer_mib = best.spacer;
    s->tuned_ms = best.k ? best.ms : 0.0;
    s->tuned_worst_ms = best.k ? worst : 0.0;
    s->tuned_candidates = best.k ? tried : 0u;
    if (glu_verbose())
        fprintf(stderr, "[glu_hip] scratch placement: value array behind a %u MiB spacer, calibration sort %.3f ms (slowest candidate %.3f ms)\n",
                best.spacer, best.ms, worst);
    return cleanup(GLU_OK);
}

// A PAIR of arrays for the caller's side of a sort, placed by measurement against a sorter whose own scratch is in place:
// the same search as tune_scratch_placement with the roles turned round -- the candidates (keys, values; the value array
// behind spacers of 0, 0.5 ... 7.5 GiB) are the arrays that are sorted, the sorter's scratch is what it is.  A pair of arrays
// is fast or slow whatever it is paired with, so a pair that sorts fast here is a good source and a good
// destination of any pass.  glu_dist_prepare places the sharded sort's send-side and receive-side arrays with it: of the four
// scatter passes of a rank's sort only the first one's source, the caller's slice, is then left to luck.  `keys` / `vals` are
// (re)allocated: `count` 4-byte elements each.  Explicit prepare calls only; silently the plain allocation when the search is
// switched off, the arrays are small or memory is short.
glu_status place_pair_by_measurement(glu_radix_sort_s* s, size_t count, Scratch& keys, Scratch& vals)
{
    const size_t bytes = count * sizeof(uint32_t);
    auto plain = [&]() -> glu_status {
        GLU_TRY(keys.reserve(bytes));
        return vals.reserve(bytes);
    };
    if (keys.size >= bytes && vals.size >= bytes) return GLU_OK;
    if (!s->tune_scratch || bytes < kTuneMinKeyBytes || s->keys.size < bytes || s->vals.size < bytes) return plain();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return plain();
    keys.release();
    vals.release();
    const size_t step_mib = 512, candidates = 16;
    if (free_b < 2 * (2 * bytes + ((candidates - 1) * step_mib << 20)) + ((size_t) 1 << 30))
    {
        if (glu_verbose()) fprintf(stderr, "[glu_hip] pair placement skipped: %zu MiB free; plain allocation\n", free_b >> 20);
        return plain();
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(GLU_ERROR_DEVICE, "hipDeviceSynchronize failed before the pair placement");
    hipStream_t st = g_dev.queue;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    {
        if (e0) (void) hipEventDestroy(e0);
        return plain();
    }
    struct Cand { void* k = nullptr; void* v = nullptr; double ms = 1e30; } best;
    double worst = 0;
    uint32_t tried = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const int was_profiling = s->profiling;
    s->profiling = 0;
    s->tuning = true; // (the sorts below must not start a search of their own)
    glu_status status = GLU_OK;
    for (size_t i = 0; i < candidates && status == GLU_OK; i++)
    {
        void *k = nullptr, *sp = nullptr, *v = nullptr;
        if (hipMalloc(&k, bytes) != hipSuccess)
        {
            (void) hipGetLastError();
            break;
        }
        if (i && hipMalloc(&sp, i * step_mib << 20) != hipSuccess)
        {
            (void) hipGetLastError();
            (void) hipFree(k);
            continue;
        }
        if (hipMalloc(&v, bytes) != hipSuccess)
        {
            (void) hipGetLastError();
            (void) hipFree(k);
            if (sp) (void) hipFree(sp);
            continue;
        }
        if (sp) (void) hipFree(sp);
        double ms = 1e30;
        for (int rep = 0; rep < 3 && status == GLU_OK; rep++) // the first run is a warm-up
        {
            hipLaunchKernelGGL(tune_fill_kernel, dim3(g_dev.num_cus * 8), dim3(256), 0, st, (uint32_t*) k, bytes / 4, 0x5EEDu + rep);
            hipLaunchKernelGGL(tune_fill_kernel, dim3(g_dev.num_cus * 8), dim3(256), 0, st, (uint32_t*) v, bytes / 4, 77u);
            (void) hipEventRecord(e0, st);
            status = sort_run<uint32_t>(s, (uint32_t*) k, (uint32_t*) v, count, 0, st);
            (void) hipEventRecord(e1, st);
            if (status == GLU_OK && hipEventSynchronize(e1) == hipSuccess && rep > 0)
            {
                float t = 0;
                if (hipEventElapsedTime(&t, e0, e1) == hipSuccess) ms = std::min(ms, (double) t);
            }
        }
        (void) hipStreamSynchronize(st);
        if (glu_verbose()) fprintf(stderr, "[glu_hip]   pair candidate: keys %p values %p (spacer %zu MiB): %.3f ms\n", k, v, i * step_mib, ms);
        if (ms < 1e29) worst = std::max(worst, ms);
        if (status == GLU_OK && ms < best.ms)
        {
            if (best.k) (void) hipFree(best.k);
            if (best.v) (void) hipFree(best.v);
            best.k = k, best.v = v, best.ms = ms;
        }
        else
        {
            (void) hipFree(k);
            (void) hipFree(v);
        }
        tried++;
        if (tried >= 8 && best.ms <= 0.93 * worst) break;
        if (std::chrono::steady_clock::now() - t_begin > std::chrono::milliseconds(1000)) break;
    }
    s->tuning = false;
    s->profiling = was_profiling;
    s->last_planned = false;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    if (best.k)
    {
        keys.ptr = best.k, keys.size = bytes;
        vals.ptr = best.v, vals.size = bytes;
        if (glu_verbose())
            fprintf(stderr, "[glu_hip] pair placement: %u candidates, calibration sort %.3f ms (slowest %.3f ms)\n", tried, best.ms, worst);
    }
    if (status != GLU_OK) return status;
    return best.k ? GLU_OK : plain();
}

// ------------------------------------------------------------------------------------------------------------
// segmented sort (radix_seg_passes.hpp): glu_radix_sort_run_segments_ptr
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSegMaxSegments = 1u << 24; // (one workgroup per segment in the scan kernel, host vectors of this length)
struct SegPiece
{
    uint64_t begin, len;
    uint32_t seg;
};

// Device image of one segmented pass's descriptors (uint32 words): subs[nsb] as (begin, end) pairs, wg_first[nwg + 1],
// seg_list[nseg + 1], seg_start[nseg].
struct SegImage
{
    std::vector<uint32_t> words;
    uint32_t nsb = 0, nwg = 0, nseg = 0, max_subs_per_wg = 0;
    size_t off_first = 0, off_list = 0, off_start = 0;
};

// Cuts the pieces (sorted by segment, in the stable order of each segment's elements) into sub-blocks: workgroup w of
// `nwg` takes the elements [w * share, (w + 1) * share) of the pieces laid end to end, a sub-block is the part of one
// piece inside one workgroup's share.  Pure host function of its arguments.
void seg_build_image(const SegPiece* pieces, size_t npieces, uint32_t nseg, const uint64_t* seg_start, uint64_t total,
                     uint32_t nwg, SegImage& img)
{
    img.nwg = nwg;
    img.nseg = nseg;
    const uint64_t share = std::max<uint64_t>(1, (total + nwg - 1) / nwg);
    std::vector<uint32_t> subs, first((size_t) nwg + 1, 0), list((size_t) nseg + 1, 0);
    subs.reserve(2 * (npieces + nwg));
    uint64_t pos = 0; // position in the pieces laid end to end
    uint32_t seg_seen = 0;
    for (size_t p = 0; p < npieces; p++)
    {
        const SegPiece& pc = pieces[p];
        if (pc.len == 0) continue;
        while (seg_seen <= pc.seg) list[seg_seen++] = (uint32_t) (subs.size() / 2); // segments [.., pc.seg] start here
        uint64_t done = 0;
        while (done < pc.len)
        {
            const uint64_t w = (pos + done) / share;
            const uint64_t room = (w + 1) * share - (pos + done);
            const uint64_t take = std::min<uint64_t>(room, pc.len - done);
            subs.push_back((uint32_t) (pc.begin + done));
            subs.push_back((uint32_t) (pc.begin + done + take));
            first[std::min<uint64_t>(w, nwg - 1) + 1]++; // (counts; turned into offsets below)
            done += take;
        }
        pos += pc.len;
    }
    img.nsb = (uint32_t) (subs.size() / 2);
    while (seg_seen <= nseg) list[seg_seen++] = img.nsb;
    img.max_subs_per_wg = 0;
    for (uint32_t w = 0; w < nwg; w++)
    {
        img.max_subs_per_wg = std::max(img.max_subs_per_wg, first[w + 1]);
        first[w + 1] += first[w];
    }
    img.words.clear();
    img.words.insert(img.words.end(), subs.begin(), subs.end());
    img.off_first = img.words.size();
    img.words.insert(img.words.end(), first.begin(), first.end());
    img.off_list = img.words.size();
    img.words.insert(img.words.end(), list.begin(), list.end());
    img.off_start = img.words.size();
    for (uint32_t g = 0; g < nseg; g++) img.words.push_back((uint32_t) seg_start[g]);
    if (img.words.size() % 2) img.words.push_back(0); // the next image's (begin, end) pairs stay 8-byte aligned
}

constexpr size_t kSegMinCount = 1 << 16; // below: gather + one sort per segment (the line kernel wants whole tiles to prefetch)
constexpr uint64_t kSegFinishMaxRuns = 1u << 18; // a segmented sort tries to end in LDS with up to this many runs (1024 segments)

// One segmented pass on the 8-bit digit at `shift`: count per sub-block, scan per segment, line scatter over sub-blocks.
// gate_mode (radix_seg_passes.hpp): kSegGateIfNot = a pass of the ordinary sequence enqueued behind an attempt to end in LDS;
// kSegGateIfFits + runs_end != 0 = the first pass of such an attempt: its count and scan kernels always run, the runs kernel
// behind them writes the run starts and the longest run (the gate), the scatter runs if that fits `gate_cap`.
glu_status launch_seg_pass(glu_radix_sort_s* s, const uint32_t* src_k, const uint32_t* src_v, uint32_t* dst_k, uint32_t* dst_v,
                           size_t count, uint32_t shift, const uint32_t* image, const SegImage& img, hipStream_t stream,
                           uint32_t gate_mode = kSegGateNone, uint32_t gate_cap = 0, uint32_t runs_end = 0)
{
    const uint32_t* gate = gate_mode != kSegGateNone ? (const uint32_t*) s->seg_gate.ptr : nullptr;
    const bool attempt = gate_mode == kSegGateIfFits;
    const uint32_t gm_count = attempt ? kSegGateNone : gate_mode; // (the attempt's count and scan make the decision)
    using G = SegLinesGeometry;
    constexpr int RADIX = 256;
    constexpr int RS = (G::KPT + 2) / 3;
    using Smem = LineSmem<uint32_t, 8, G::THREADS, G::KPT, true>;
    auto scatter_nt = radix_scatter_lines_kernel<uint32_t, 8, G::THREADS, G::KPT, false, true, 0, false, RS, true, true, 0, true>;
    auto scatter_plain = radix_scatter_lines_kernel<uint32_t, 8, G::THREADS, G::KPT, false, true, 0, false, RS, true, false, 0, true>;
    static std::once_flag lds_opt_in;
    static hipError_t lds_opt_in_result = hipSuccess;
    std::call_once(lds_opt_in, [&] {
        lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_nt, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
        if (lds_opt_in_result == hipSuccess)
            lds_opt_in_result = hipFuncSetAttribute((const void*) scatter_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem));
    });
    HIP_TRY(lds_opt_in_result);
    uint32_t* table = (uint32_t*) s->table.ptr;
    const uint2* subs = (const uint2*) image;
    s->mark(stream);
    hipLaunchKernelGGL((radix_seg_count_kernel<uint32_t, 8, 1024>), dim3(img.nwg), dim3(1024), 0, stream, src_k, subs, image + img.off_first, table,
                       shift, 255u, gate, gate_cap, gm_count, (const uint32_t*) nullptr, (const PassPlan*) nullptr, 0u, 0u);
    HIP_TRY(hipGetLastError());
    s->mark(stream);
    hipLaunchKernelGGL((radix_seg_scan_kernel<RADIX>), dim3(img.nseg), dim3(RADIX), 0, stream, table, image + img.off_list,
                       image + img.off_start, gate, gate_cap, gm_count, (const uint32_t*) nullptr);
    HIP_TRY(hipGetLastError());
    if (attempt)
    {
        hipLaunchKernelGGL((radix_seg_runs_kernel<RADIX>), dim3(1), dim3(1024), 0, stream, (const uint32_t*) table, image + img.off_list,
                           image + img.off_start, img.nseg, runs_end, (uint32_t*) s->finish_starts.ptr, (uint32_t*) s->seg_gate.ptr);
        HIP_TRY(hipGetLastError());
    }
    s->mark(stream, true);
    hipLaunchKernelGGL(s->nt_stores ? scatter_nt : scatter_plain, dim3(img.nwg), dim3(G::THREADS), sizeof(Smem), stream, src_k, src_v,
                       dst_k, dst_v, (const uint32_t*) table, (const uint32_t*) nullptr, (uint32_t) count, shift, 255u, 0u,
                       (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, subs, 0u, image + img.off_first, gate, gate_cap, gate_mode, (const uint32_t*) nullptr);
    HIP_TRY(hipGetLastError());
    s->mark(stream, true);
    return GLU_OK;
}

// Host half of a segmented sort: the pieces sorted by segment (stably: the caller's order of a segment's pieces is the
// order of its elements), where every segment starts in the output, and the descriptor images of the passes.  Nothing is
// enqueued, so a caller (glu_dist) can look at `max_subs_per_wg` -- a workgroup pays about two tiles' time per sub-block,
// whatever its size -- and decide for another way before anything runs.
struct SegPlan
{
    std::vector<SegPiece> pieces;
    std::vector<uint64_t> seg_start; // [nseg + 1]
    uint32_t nseg = 0, passes = 0, bits = 0;
    size_t count = 0;
    bool by_copies = false; // small or unaligned inputs: gather with copies + one ordinary sort per segment
    SegImage first, later;
    uint32_t max_subs_per_wg() const { return std::max(first.max_subs_per_wg, later.max_subs_per_wg); }
};

// `origin`: where the first segment starts in the output (and in the arrays of the intermediate passes): the segments of a
// plan occupy [origin, origin + count) there, so several plans can share one set of (aligned) arrays -- the groups of a
// sharded sort's exchange in rounds, glu_dist_impl.hpp.
void seg_make_plan(glu_radix_sort_s* s, std::vector<SegPiece>&& pieces, uint32_t nseg, size_t count, uint32_t bits, bool aligned,
                   SegPlan& plan, uint64_t origin = 0)
{
    plan.pieces = std::move(pieces);
    std::stable_sort(plan.pieces.begin(), plan.pieces.end(), [](const SegPiece& a, const SegPiece& b) { return a.seg < b.seg; });
    plan.nseg = nseg;
    plan.count = count;
    plan.bits = bits;
    plan.passes = bits / 8;
    plan.seg_start.assign((size_t) nseg + 1, 0);
    for (const SegPiece& pc : plan.pieces) plan.seg_start[pc.seg + 1] += pc.len;
    plan.seg_start[0] = origin;
    for (uint32_t g = 0; g < nseg; g++) plan.seg_start[g + 1] += plan.seg_start[g];
    plan.by_copies = count < kSegMinCount || !aligned || plan.passes == 0;
    if (plan.by_copies) return;
    // descriptors of the first pass (the caller's pieces) and of the later ones (whole segments of the pass before)
    const uint32_t nwg = usable_cus(s);
    seg_build_image(plan.pieces.data(), plan.pieces.size(), nseg, plan.seg_start.data(), count, nwg, plan.first);
    if (plan.passes > 1)
    {
        std::vector<SegPiece> whole(nseg);
        for (uint32_t g = 0; g < nseg; g++) whole[g] = SegPiece{plan.seg_start[g], plan.seg_start[g + 1] - plan.seg_start[g], g};
        seg_build_image(whole.data(), whole.size(), nseg, plan.seg_start.data(), count, nwg, plan.later);
    }
}

// Device half: enqueues the passes of `plan` on `stream`.
// Will a segmented sort by this plan try to end in LDS (given a third pair of arrays to land its counting pass in)?  The tile and,
// for long runs, the number of workgroups a run is split over follow from the run length uniformly drawn keys would give in the
// LARGEST segment (segments may be of any sizes, empty ones included, and the host knows them).  Also asked by glu_dist, which
// allocates the landing pair of the exchange only for sorts that will use it.
bool seg_finish_choice_for(const glu_radix_sort_s* s, uint32_t passes, uint32_t bits, uint64_t nruns, uint64_t largest, uint32_t& geo,
                           uint32_t& split_log2)
{
    geo = 0, split_log2 = 0;
    if (!(s->seg_finish && s->lds_finish && passes >= 2 && bits == passes * 8 && nruns <= kSegFinishMaxRuns)) return false;
    split_log2 = std::min(s->seg_split_min, s->seg_split_max);
    for (; split_log2 <= s->seg_split_max && !geo; split_log2++)
    {
        geo = finish_geometry_for((size_t) (largest >> split_log2), 256u, std::min(s->seg_max_geo, kSegFinishGeometries));
        if (split_log2 > 0 && geo > s->seg_split_geo) geo = 0; // (split runs take the tiles that share a CU four at a time)
    }
    if (!geo) return false;
    split_log2--;
    return true;
}
bool seg_finish_choice(const glu_radix_sort_s* s, const SegPlan& plan, uint32_t& geo, uint32_t& split_log2)
{
    uint64_t largest = 0;
    for (uint32_t g = 0; g < plan.nseg; g++) largest = std::max<uint64_t>(largest, plan.seg_start[g + 1] - plan.seg_start[g]);
    return seg_finish_choice_for(s, plan.passes, plan.bits, (uint64_t) plan.nseg * 256u, largest, geo, split_log2);
}

glu_status seg_run_plan(glu_radix_sort_s* s, const SegPlan& plan, uint32_t* in_k, uint32_t* in_v, uint32_t* out_k, uint32_t* out_v,
                        hipStream_t stream)
{
    const size_t count = plan.count;
    const uint32_t passes = plan.passes, nseg = plan.nseg;
    s->cur_kind = 0, s->cur_behind = false, s->cur_seq = 0; // (whatever an earlier call that failed half-way left)
    // (the scratch arrays are indexed like `out`: a plan with an origin -- a group of a sharded sort's rounds -- reaches further)
    GLU_TRY(sort_prepare(s, (plan.seg_start.empty() ? 0 : (size_t) plan.seg_start[0]) + count, sizeof(uint32_t), true));
    if (plan.by_copies)
    {
        // small (or unaligned, or nothing to sort by): lay the segments out with copies, then one sort per segment
        uint64_t at = plan.seg_start.empty() ? 0 : plan.seg_start[0];
        for (const SegPiece& pc : plan.pieces)
        {
            if (pc.len == 0) continue;
            HIP_TRY(hipMemcpyAsync(out_k + at, in_k + pc.begin, pc.len * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(out_v + at, in_v + pc.begin, pc.len * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            at += pc.len;
        }
        if (passes)
            for (uint32_t g = 0; g < nseg; g++)
            {
                const uint64_t len = plan.seg_start[g + 1] - plan.seg_start[g];
                if (len > 1)
                    GLU_TRY(sort_bits<uint32_t>(s, out_k + plan.seg_start[g], out_v + plan.seg_start[g], (size_t) len, 0u, plan.bits, stream));
            }
        return GLU_OK;
    }
    uint32_t* tmp_k = (uint32_t*) s->keys.ptr;
    uint32_t* tmp_v = (uint32_t*) s->vals.ptr;
    const SegImage &first = plan.first, &later = plan.later;
    const size_t bytes = (first.words.size() + later.words.size()) * sizeof(uint32_t);
    GLU_TRY(s->seg_desc.reserve(std::max<size_t>(bytes, 1 << 16)));
    GLU_TRY(s->table.reserve((size_t) std::max(first.nsb, later.nsb) * 256 * sizeof(uint32_t)));
    glu_radix_sort_s::SegStage& st = s->seg_stage[s->seg_stage_next++ % 16];
    if (st.in_flight) HIP_TRY(hipEventSynchronize(st.copied)); // (sixteen calls ago: long done)
    if (st.size < bytes)
    {
        if (st.host) HIP_TRY(hipHostFree(st.host));
        st.host = nullptr;
        st.size = 0;
        HIP_TRY(hipHostMalloc(&st.host, std::max<size_t>(bytes, 1 << 16)));
        st.size = std::max<size_t>(bytes, 1 << 16);
    }
    if (!st.copied) HIP_TRY(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
    memcpy(st.host, first.words.data(), first.words.size() * sizeof(uint32_t));
    if (!later.words.empty())
        memcpy((uint32_t*) st.host + first.words.size(), later.words.data(), later.words.size() * sizeof(uint32_t));
    HIP_TRY(hipMemcpyAsync(s->seg_desc.ptr, st.host, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(st.copied, stream));
    st.in_flight = true;
    const uint32_t* image_first = (const uint32_t*) s->seg_desc.ptr;
    const uint32_t* image_later = image_first + first.words.size();

    // A segmented sort that ENDS IN LDS (radix_seg_passes.hpp, radix_lds_finish.hpp): ONE counting pass, on the top digit of the
    // bits to sort by, straight from the caller's pieces into the sort's own scratch arrays -- after it the array is nseg x 256
    // runs (segment, top digit) whose starts are in the pass's scanned table -- and one pass that orders every run by the
    // remaining low bits inside LDS on its way into `out`: 20 + 16 B per pair instead of passes x 20.  The tile and, for long
    // runs, the number of workgroups a run is split over follow from the run length uniformly drawn keys would give (mean + 6
    // sigma); runs that come out longer are walked in pieces by radix_finish_ranges_kernel.  The device decides by the longest
    // run (the gate) before the pass's scatter moves anything: beyond 32 tiles per workgroup the ordinary passes, enqueued
    // behind, run instead (they return at once otherwise).  Needs the scratch arrays to be a third pair (in != scratch != out).
    uint32_t gate_mode = kSegGateNone, gate_cap = 0;
    const uint64_t nruns = (uint64_t) nseg * 256u;
    s->last_seg_finish_attempted = false;
    const bool third_pair = tmp_k != in_k && tmp_k != out_k && tmp_v != in_v && tmp_v != out_v;
    {
        uint32_t geo = 0, split_log2 = 0;
        if (third_pair && seg_finish_choice(s, plan, geo, split_log2))
        {
            GLU_TRY(s->finish_starts.reserve(((size_t) std::max<uint64_t>(nruns, kFinishRuns) + 1) * sizeof(uint32_t)));
            GLU_TRY(s->seg_gate.reserve(64));
            gate_cap = (uint32_t) std::min<uint64_t>(0xFFFFFFFFull, (uint64_t) finish_geometry_capacity(geo) * 32u << split_log2);
            const uint32_t end = (uint32_t) (plan.seg_start[0] + count);
            s->cur_kind = 3;
            GLU_TRY(launch_seg_pass(s, in_k, in_v, tmp_k, tmp_v, count, plan.bits - 8, image_first, first, stream, kSegGateIfFits, gate_cap, end));
            s->cur_kind = 2;
            s->mark(stream);
            s->mark(stream);
            s->mark(stream, true);
            GLU_TRY(launch_seg_finish(tmp_k, tmp_v, out_k, out_v, (const uint32_t*) s->finish_starts.ptr, (uint32_t) nruns, geo, split_log2,
                                      plan.bits - 8, (const uint32_t*) s->seg_gate.ptr, gate_cap, s->finish_rank_bits, stream));
            s->mark(stream, true);
            s->cur_kind = 0;
            gate_mode = kSegGateIfNot;
            s->last_seg_finish_attempted = true;
            s->last_seg_finish_capacity = gate_cap;
            s->last_seg_finish_runs = (uint32_t) nruns;
            s->last_seg_finish_tile = finish_geometry_capacity(geo);
            s->last_seg_finish_split = 1u << split_log2;
        }
    }

    // the arrays of every pass: the last one writes `out`; with an odd number of passes they alternate in -> out -> in -> out,
    // with an even number the sort's own scratch stands in for `out` until the last pass (in -> tmp -> in -> tmp -> out)
    const uint32_t* src_k = in_k;
    const uint32_t* src_v = in_v;
    for (uint32_t p = 0; p < passes; p++)
    {
        const bool last = p + 1 == passes;
        uint32_t* dst_k = last ? out_k : (src_k == in_k ? ((passes & 1u) ? out_k : tmp_k) : in_k);
        uint32_t* dst_v = last ? out_v : (src_v == in_v ? ((passes & 1u) ? out_v : tmp_v) : in_v);
        s->cur_kind = gate_mode == kSegGateIfNot ? 4 : 0;
        s->cur_behind = gate_mode == kSegGateIfNot;
        GLU_TRY(launch_seg_pass(s, src_k, src_v, dst_k, dst_v, count, p * 8, p == 0 ? image_first : image_later,
                                p == 0 ? first : later, stream, gate_mode, gate_cap));
        src_k = dst_k;
        src_v = dst_v;
    }
    s->cur_kind = 0;
    s->cur_behind = false;
    return GLU_OK;
}
} // namespace

// The switches of a sort object (tests, tuning, A/B runs): one table for glu_radix_sort_set_option and for the environment
// defaults glu_radix_sort_create reads (GLU_HIP_<NAME>).  Every field is documented where glu_radix_sort_s declares it.
namespace
{
struct SortOption
{
    const char* name;
    void (*set)(glu_radix_sort_s*, long long);
};
#define GLU_OPT(NAME, ...) {NAME, [](glu_radix_sort_s* s, long long v) { (void) v; __VA_ARGS__; }}
const SortOption kSortOptions[] = {
    GLU_OPT("DIGIT_BITS", if (v == 4 || v == 8) s->digit_bits = (uint32_t) v),
    GLU_OPT("SORT_BLOCKS", if (v > 0) s->max_blocks = (uint32_t) v),
    GLU_OPT("SORT_SMALL", s->force_small = v != 0),
    GLU_OPT("SORT_NO_SINGLE_BLOCK", s->no_single_block = v != 0),
    GLU_OPT("SORT_NO_FUSED_SCAN", s->no_fused_scan = v != 0),
    GLU_OPT("SORT_NO_PLAN", s->no_plan = v != 0),
    GLU_OPT("SORT_NO_LINES", s->no_lines = v != 0),
    GLU_OPT("SCRATCH_TUNE", s->tune_scratch = v != 0),
    GLU_OPT("SORT_PAIRS", s->pairs = v != 0),
    GLU_OPT("SORT_EQUAL_SHARES", s->equal_shares = v != 0),
    GLU_OPT("SORT_NO_BIT_SHORTCUT", s->no_bit_shortcut = v != 0),
    GLU_OPT("SORT_PAIR_MIN", s->pair_min = (size_t) v),
    GLU_OPT("SORT_LDS_FINISH", s->lds_finish = v != 0),
    GLU_OPT("SEG_LDS_FINISH", s->seg_finish = v != 0),
    GLU_OPT("SORT_LONG_RUNS", s->long_runs = v != 0),
    GLU_OPT("SORT_DEVICE_TOP", s->device_top = v != 0),
    GLU_OPT("SORT_FORK", s->fork_behind = v != 0),
    GLU_OPT("SEG_SPLIT_MAX", s->seg_split_max = (uint32_t) std::min<long long>(std::max<long long>(v, 0), 3)),
    GLU_OPT("FINISH_RANK_BITS", if (v >= 8) s->finish_rank_bits = (uint32_t) v),
    GLU_OPT("SEG_SPLIT_MIN", s->seg_split_min = (uint32_t) std::min<long long>(std::max<long long>(v, 0), 3)),
    GLU_OPT("SEG_MAX_GEO", s->seg_max_geo = (uint32_t) std::min<long long>(std::max<long long>(v, 1), 5)),
    GLU_OPT("SEG_SPLIT_GEO", s->seg_split_geo = (uint32_t) std::min<long long>(std::max<long long>(v, 1), 4)),
    GLU_OPT("SORT_FINISH_MIN", s->finish_min = (size_t) v),
    GLU_OPT("SORT_FINISH_BACKOFF", s->finish_backoff = (uint32_t) std::max<long long>(0, v)),
    GLU_OPT("SORT_PAIR_UNIT_DIV", if (v >= 0) s->pair_unit_div = (uint32_t) v), // 0: no limit
    GLU_OPT("SORT_NT_STORES", s->nt_stores = v != 0),
    GLU_OPT("SORT_NT_MIN_BYTES", s->nt_min_bytes = (size_t) v),
    GLU_OPT("SORT_LARGE_MIN", s->large_min = (size_t) v),
};
#undef GLU_OPT
}

extern "C" {

glu_status glu_radix_sort_create(glu_radix_sort* out)
{
    GLU_TRY(enter());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    glu_radix_sort_s* s = new glu_radix_sort_s();
    // the process environment as DEFAULTS for new objects (GLU_HIP_<OPTION>=value, read here, when an object is created);
    // programs and tests set options on the object: glu_radix_sort_set_option
    for (const SortOption& o : kSortOptions)
    {
        char name[96];
        snprintf(name, sizeof(name), "GLU_HIP_%s", o.name);
        if (const char* e = glu_env(name)) o.set(s, atoll(e));
    }
    *out = s;
    return GLU_OK;
}

glu_status glu_radix_sort_destroy(glu_radix_sort sort)
{
    GLU_TRY(enter());
    if (!sort) return GLU_OK;
    // the object may last have been used on a caller's stream (the *_ptr entry points): drain the device, not only the
    // library queue, before its scratch goes away (RAII of the reference: RadixSort.hpp:194-200, gl_utils.hpp:184-188)
    (void) hipDeviceSynchronize();
    for (Scratch* sc : {&sort->keys, &sort->vals, &sort->table, &sort->plan, &sort->pair_t2, &sort->pair_table, &sort->pair_ranges,
                        &sort->pair_sub, &sort->pair_wide, &sort->seg_desc, &sort->seg_zero, &sort->finish_lengths, &sort->finish_starts, &sort->finish_crowded, &sort->finish_outcomes, &sort->seg_gate, &sort->long_image, &sort->long_hdr, &sort->long_bits})
        sc->release();
    if (sort->finish_hint) (void) hipHostFree(sort->finish_hint);
    for (hipEvent_t e : {sort->ev_fork, sort->ev_unit, sort->ev_fork2, sort->ev_join})
        if (e) (void) hipEventDestroy(e);
    if (sort->side) (void) hipStreamDestroy(sort->side);
    for (hipEvent_t e : sort->events) (void) hipEventDestroy(e);
    for (glu_radix_sort_s::SegStage& st : sort->seg_stage)
    {
        if (st.host) (void) hipHostFree(st.host);
        if (st.copied) (void) hipEventDestroy(st.copied);
    }
    delete sort;
    return GLU_OK;
}

glu_status glu_radix_sort_prepare(glu_radix_sort sort, size_t count)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    return sort_prepare(sort, count, sizeof(uint32_t), true, true);
}

glu_status glu_radix_sort_prepare_ex(glu_radix_sort sort, size_t count, size_t key_bytes, int with_vals)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (key_bytes != 4 && key_bytes != 8) return fail(GLU_ERROR_INVALID_ARGUMENT, "key_bytes must be 4 or 8 (got %zu)", key_bytes);
    return sort_prepare(sort, count, key_bytes, with_vals != 0, true);
}

glu_status glu_radix_sort_prepare_u64(glu_radix_sort sort, size_t count)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    return sort_prepare(sort, count, sizeof(uint64_t), true, true);
}

glu_status glu_radix_sort_run_ptr(glu_radix_sort sort, uint32_t* keys, uint32_t* vals, size_t count, size_t num_steps,
                                  void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (count == 0) return GLU_OK; // an empty shard: nothing to sort, NULL arrays are fine (torch gives data_ptr() == 0)
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    if (!vals) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid value buffer");
    return sort_run<uint32_t>(sort, keys, vals, count, num_steps, pick_stream(stream));
}

glu_status glu_radix_sort_run_u64_ptr(glu_radix_sort sort, uint64_t* keys, uint32_t* vals, size_t count,
                                      size_t num_steps, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (count == 0) return GLU_OK;
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    if (!vals) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid value buffer");
    return sort_run<uint64_t>(sort, keys, vals, count, num_steps, pick_stream(stream));
}

glu_status glu_radix_sort_run(glu_radix_sort sort, glu_buffer key_buffer, glu_buffer val_buffer, size_t count,
                              size_t num_steps)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    Buffer k, v;
    GLU_TRY(lookup(key_buffer, k, "key buffer"));   // RadixSort.hpp:275
    GLU_TRY(lookup(val_buffer, v, "value buffer")); // RadixSort.hpp:276
    if (count > 1 && (k.size / sizeof(uint32_t) < count || v.size / sizeof(uint32_t) < count))
        return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu exceeds the key/value buffer size", count);
    return sort_run<uint32_t>(sort, (uint32_t*) k.ptr, (uint32_t*) v.ptr, count, num_steps, g_dev.queue);
}

glu_status glu_radix_sort_run_u64(glu_radix_sort sort, glu_buffer key_buffer, glu_buffer val_buffer, size_t count,
                                  size_t num_steps)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    Buffer k, v;
    GLU_TRY(lookup(key_buffer, k, "key buffer"));
    GLU_TRY(lookup(val_buffer, v, "value buffer"));
    if (count > 1 && (k.size / sizeof(uint64_t) < count || v.size / sizeof(uint32_t) < count))
        return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu exceeds the key/value buffer size", count);
    return sort_run<uint64_t>(sort, (uint64_t*) k.ptr, (uint32_t*) v.ptr, count, num_steps, g_dev.queue);
}

glu_status glu_radix_sort_run_keys_ptr(glu_radix_sort sort, uint32_t* keys, size_t count, size_t num_steps, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (count == 0) return GLU_OK; // an empty shard: NULL arrays are fine
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    return sort_run<uint32_t>(sort, keys, nullptr, count, num_steps, pick_stream(stream));
}

glu_status glu_radix_sort_run_keys_u64_ptr(glu_radix_sort sort, uint64_t* keys, size_t count, size_t num_steps, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (count == 0) return GLU_OK;
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    return sort_run<uint64_t>(sort, keys, nullptr, count, num_steps, pick_stream(stream));
}

glu_status glu_radix_sort_run_typed_ptr(glu_radix_sort sort, void* keys, uint32_t* vals, size_t count, glu_key_type key_type,
                                        void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if ((int) key_type < (int) GLU_KEY_UINT32 || (int) key_type > (int) GLU_KEY_FLOAT64)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key type: %d", (int) key_type);
    if (count == 0) return GLU_OK;
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    hipStream_t st = pick_stream(stream);
    switch (key_type)
    {
    case GLU_KEY_UINT32: return sort_run<uint32_t>(sort, (uint32_t*) keys, vals, count, 0, st, KEY_XF_NONE);
    case GLU_KEY_INT32: return sort_run<uint32_t>(sort, (uint32_t*) keys, vals, count, 0, st, KEY_XF_SIGNED);
    case GLU_KEY_FLOAT32: return sort_run<uint32_t>(sort, (uint32_t*) keys, vals, count, 0, st, KEY_XF_FLOAT);
    case GLU_KEY_UINT64: return sort_run<uint64_t>(sort, (uint64_t*) keys, vals, count, 0, st, KEY_XF_NONE);
    case GLU_KEY_INT64: return sort_run<uint64_t>(sort, (uint64_t*) keys, vals, count, 0, st, KEY_XF_SIGNED);
    case GLU_KEY_FLOAT64: return sort_run<uint64_t>(sort, (uint64_t*) keys, vals, count, 0, st, KEY_XF_FLOAT);
    default: return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key type: %d", (int) key_type);
    }
}

glu_status glu_radix_sort_run_bit_range_ptr(glu_radix_sort sort, void* keys, uint32_t* vals, size_t count, uint32_t key_bits,
                                            uint32_t begin_bit, uint32_t end_bit, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (key_bits != 32 && key_bits != 64) return fail(GLU_ERROR_INVALID_ARGUMENT, "key_bits must be 32 or 64 (got %u)", key_bits);
    if (begin_bit > end_bit || end_bit > key_bits)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "bad bit range [%u, %u) for %u-bit keys", begin_bit, end_bit, key_bits);
    if (count == 0) return GLU_OK;
    if (!keys) return fail(GLU_ERROR_INVALID_ARGUMENT, "Invalid key buffer");
    hipStream_t st = pick_stream(stream);
    if (key_bits == 32) return sort_bits<uint32_t>(sort, (uint32_t*) keys, vals, count, begin_bit, end_bit, st);
    return sort_bits<uint64_t>(sort, (uint64_t*) keys, vals, count, begin_bit, end_bit, st);
}

glu_status glu_radix_sort_run_keys(glu_radix_sort sort, glu_buffer key_buffer, size_t count, size_t num_steps)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    Buffer k;
    GLU_TRY(lookup(key_buffer, k, "key buffer"));
    if (count > 1 && k.size / sizeof(uint32_t) < count)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu exceeds the key buffer size", count);
    return sort_run<uint32_t>(sort, (uint32_t*) k.ptr, nullptr, count, num_steps, g_dev.queue);
}

glu_status glu_radix_sort_partition_ptr(glu_radix_sort sort, const uint32_t* src_keys, const uint32_t* src_vals,
                                        uint32_t* dst_keys, uint32_t* dst_vals, size_t count, uint32_t shift,
                                        uint32_t bits, uint32_t* digit_histogram, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (count > 0)
    {
        if (!src_keys || !src_vals || !dst_keys || !dst_vals) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL array");
        if (src_keys == dst_keys || src_vals == dst_vals)
            return fail(GLU_ERROR_INVALID_ARGUMENT, "partition needs distinct source and destination");
    }
    if (bits < 1 || bits > 8 || shift + bits > 32) return fail(GLU_ERROR_INVALID_ARGUMENT, "bad digit: shift %u bits %u", shift, bits);
    if (count > 0xFFFF0000ull) return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu does not fit 32-bit indexing", count);
    hipStream_t st = pick_stream(stream);
    if (count == 0)
    {
        if (digit_histogram) HIP_TRY(hipMemsetAsync(digit_histogram, 0, ((size_t) 1 << bits) * 4, st));
        return GLU_OK;
    }
    // table only (no key/val scratch needed)
    uint64_t cap = (uint64_t) g_dev.num_cus * kMaxBlocksPerCu;
    GLU_TRY(sort->table.reserve((kMaxRadix * cap + kMaxRadix) * sizeof(uint32_t)));
    return dispatch_pass<uint32_t>(sort, src_keys, src_vals, dst_keys, dst_vals, count, shift, bits, digit_histogram, st);
}

glu_status glu_radix_sort_run_segments_ptr(glu_radix_sort sort, uint32_t* in_keys, uint32_t* in_vals, uint32_t* out_keys,
                                           uint32_t* out_vals, size_t count, const uint64_t* piece_begin,
                                           const uint64_t* piece_len, const uint32_t* piece_segment, size_t num_pieces,
                                           uint32_t num_segments, uint32_t key_bits, void* stream)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (key_bits > 32 || key_bits % 8 != 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "key_bits must be 0, 8, 16, 24 or 32 (got %u)", key_bits);
    if (count > 0xFFFF0000ull) return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu does not fit 32-bit indexing", count);
    if (num_pieces > 0 && (!piece_begin || !piece_len || !piece_segment)) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL piece array");
    if (num_segments == 0 && num_pieces > 0) return fail(GLU_ERROR_INVALID_ARGUMENT, "pieces but no segments");
    if (num_segments > kSegMaxSegments) return fail(GLU_ERROR_INVALID_ARGUMENT, "num_segments %u exceeds %u", num_segments, kSegMaxSegments);
    std::vector<SegPiece> pieces(num_pieces);
    uint64_t total = 0;
    for (size_t i = 0; i < num_pieces; i++)
    {
        if (piece_segment[i] >= num_segments) return fail(GLU_ERROR_INVALID_ARGUMENT, "piece %zu names segment %u of %u", i, piece_segment[i], num_segments);
        if (piece_begin[i] > count || piece_len[i] > count - piece_begin[i])
            return fail(GLU_ERROR_INVALID_ARGUMENT, "piece %zu [%llu, +%llu) exceeds count %zu", i, (unsigned long long) piece_begin[i],
                        (unsigned long long) piece_len[i], count);
        pieces[i] = SegPiece{piece_begin[i], piece_len[i], piece_segment[i]};
        total += piece_len[i];
    }
    if (total != count) return fail(GLU_ERROR_INVALID_ARGUMENT, "the pieces hold %llu elements, count is %zu", (unsigned long long) total, count);
    if (count == 0) return GLU_OK;
    {
        // the pieces must tile [0, count): together they hold count elements, so it is enough that none overlaps another
        std::vector<std::pair<uint64_t, uint64_t>> spans;
        spans.reserve(num_pieces);
        for (const SegPiece& pc : pieces)
            if (pc.len) spans.emplace_back(pc.begin, pc.len);
        std::sort(spans.begin(), spans.end());
        uint64_t end = 0;
        for (const auto& sp : spans)
        {
            if (sp.first != end)
                return fail(GLU_ERROR_INVALID_ARGUMENT, "the pieces do not tile the input: element %llu is in %s piece",
                            (unsigned long long) std::min(sp.first, end), sp.first < end ? "more than one" : "no");
            end = sp.first + sp.second;
        }
    }
    if (!in_keys || !in_vals || !out_keys || !out_vals) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL array");
    if (in_keys == out_keys || in_vals == out_vals) return fail(GLU_ERROR_INVALID_ARGUMENT, "the segmented sort needs distinct input and output arrays");
    // the sub-block descriptors of a call travel through a ring of pinned staging buffers that later calls overwrite, and a
    // prepared sorter may wait for a staging buffer's previous copy: a captured graph would replay neither correctly
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void) hipStreamIsCapturing(pick_stream(stream), &capturing);
    if (capturing != hipStreamCaptureStatusNone)
        return fail(GLU_ERROR_INVALID_STATE, "glu_radix_sort_run_segments_ptr cannot be captured into a graph (its descriptors are staged per call)");
    const bool aligned = (((uintptr_t) in_keys | (uintptr_t) in_vals | (uintptr_t) out_keys | (uintptr_t) out_vals) & 15u) == 0;
    SegPlan plan;
    seg_make_plan(sort, std::move(pieces), num_segments, count, key_bits, aligned, plan);
    return seg_run_plan(sort, plan, in_keys, in_vals, out_keys, out_vals, pick_stream(stream));
}

glu_status glu_radix_sort_plan_segments(const uint64_t* piece_begin, const uint64_t* piece_len, const uint32_t* piece_segment,
                                        size_t num_pieces, uint32_t num_segments, uint32_t num_workgroups, uint32_t* sub_blocks,
                                        size_t sub_block_capacity, uint32_t* workgroup_first, uint32_t* segment_first,
                                        uint64_t* segment_start, size_t* num_sub_blocks)
{
    // host only (no device needed): the cut of a segmented pass into sub-blocks, for inspection and tests
    if (num_pieces > 0 && (!piece_begin || !piece_len || !piece_segment)) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL piece array");
    if (num_workgroups == 0 || !num_sub_blocks) return fail(GLU_ERROR_INVALID_ARGUMENT, "bad arguments");
    if (num_segments > kSegMaxSegments) return fail(GLU_ERROR_INVALID_ARGUMENT, "num_segments %u exceeds %u", num_segments, kSegMaxSegments);
    if (num_workgroups > (1u << 20)) return fail(GLU_ERROR_INVALID_ARGUMENT, "num_workgroups %u exceeds %u", num_workgroups, 1u << 20);
    std::vector<SegPiece> pieces(num_pieces);
    uint64_t total = 0;
    for (size_t i = 0; i < num_pieces; i++)
    {
        if (piece_segment[i] >= num_segments) return fail(GLU_ERROR_INVALID_ARGUMENT, "piece %zu names segment %u of %u", i, piece_segment[i], num_segments);
        pieces[i] = SegPiece{piece_begin[i], piece_len[i], piece_segment[i]};
        total += piece_len[i];
    }
    if (total > 0xFFFF0000ull) return fail(GLU_ERROR_INVALID_ARGUMENT, "the pieces hold %llu elements: more than 32-bit indexing", (unsigned long long) total);
    std::stable_sort(pieces.begin(), pieces.end(), [](const SegPiece& a, const SegPiece& b) { return a.seg < b.seg; });
    std::vector<uint64_t> start((size_t) num_segments + 1, 0);
    for (const SegPiece& pc : pieces) start[pc.seg + 1] += pc.len;
    for (uint32_t g = 0; g < num_segments; g++) start[g + 1] += start[g];
    SegImage img;
    seg_build_image(pieces.data(), pieces.size(), num_segments, start.data(), total, num_workgroups, img);
    *num_sub_blocks = img.nsb;
    if (img.nsb > sub_block_capacity) return fail(GLU_ERROR_INVALID_ARGUMENT, "%u sub-blocks, room for %zu", img.nsb, sub_block_capacity);
    if (sub_blocks) memcpy(sub_blocks, img.words.data(), (size_t) img.nsb * 2 * sizeof(uint32_t));
    if (workgroup_first) memcpy(workgroup_first, img.words.data() + img.off_first, ((size_t) num_workgroups + 1) * sizeof(uint32_t));
    if (segment_first) memcpy(segment_first, img.words.data() + img.off_list, ((size_t) num_segments + 1) * sizeof(uint32_t));
    if (segment_start)
        for (uint32_t g = 0; g <= num_segments; g++) segment_start[g] = start[g];
    return GLU_OK;
}

glu_status glu_radix_sort_set_digit_bits(glu_radix_sort sort, uint32_t bits)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (bits != 4 && bits != 8) return fail(GLU_ERROR_INVALID_ARGUMENT, "digit bits must be 4 or 8 (got %u)", bits);
    sort->digit_bits = bits;
    return GLU_OK;
}

glu_status glu_radix_sort_set_option(glu_radix_sort sort, const char* name, long long value)
{
    GLU_TRY(enter());
    if (!sort || !name) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort / name is NULL");
    if (strncasecmp(name, "GLU_HIP_", 8) == 0) name += 8; // (the environment's spelling is accepted)
    for (const SortOption& o : kSortOptions)
        if (strcasecmp(name, o.name) == 0)
        {
            o.set(sort, value);
            return GLU_OK;
        }
    return fail(GLU_ERROR_INVALID_ARGUMENT, "no such option: %s", name);
}

glu_status glu_radix_sort_get_digit_bits(glu_radix_sort sort, uint32_t* bits)
{
    GLU_TRY(enter());
    if (!sort || !bits) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *bits = sort->digit_bits;
    return GLU_OK;
}

glu_status glu_radix_sort_set_profiling(glu_radix_sort sort, int enable)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    sort->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    if (!enable) sort->events_used = 0, sort->slots.clear(), sort->pass_kinds.clear(), sort->pass_seqs.clear();
    return GLU_OK;
}

glu_status glu_radix_sort_read_plan(glu_radix_sort sort, uint32_t* skipped, uint32_t* counted_alone, uint32_t* pair_role, size_t passes)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (passes > (size_t) kPlanMaxPasses) return fail(GLU_ERROR_INVALID_ARGUMENT, "at most %d passes", kPlanMaxPasses);
    if (!sort->plan.ptr) return fail(GLU_ERROR_INVALID_STATE, "no sort has run on this object");
    PassPlan host;
    memset(&host, 0, sizeof(host));
    // a sort below 2^22 elements runs without a device-side plan: every pass ran, none was paired
    if (sort->last_planned) HIP_TRY(hipMemcpy(&host, sort->plan.ptr, sizeof(host), hipMemcpyDeviceToHost));
    // a sort that tried to end in LDS has its two top-bit passes in front of the ordinary ones, which are what this call
    // reports (glu_radix_sort_read_finish tells which of the two sequences ran)
    const size_t first = sort->last_planned && sort->last_finish_attempted ? 2 : 0;
    for (size_t p = 0; p < passes; p++)
    {
        const bool have = first + p < (size_t) kPlanMaxPasses;
        if (skipped) skipped[p] = have ? host.skip[first + p] : 0u;
        if (counted_alone) counted_alone[p] = have ? host.pair_fallback[first + p] : 0u;
        if (pair_role) pair_role[p] = sort->last_planned && have ? sort->last_pair_roles[first + p] : 0u;
    }
    return GLU_OK;
}

glu_status glu_radix_sort_plan_finish(size_t count, uint32_t key_bytes, uint32_t* first_capacity, uint32_t* last_capacity)
{
    // host only (no device needed): which tiles of the in-LDS pass a whole-key sort of `count` elements would enqueue by default
    if (key_bytes != 4 && key_bytes != 8) return fail(GLU_ERROR_INVALID_ARGUMENT, "key_bytes must be 4 or 8 (got %u)", key_bytes);
    uint32_t first = 0, last = 0;
    if (count >= finish_min_count(key_bytes) && count >= kPlanMinCount && count <= 0xFFFF0000ull)
    {
        first = finish_geometry_for(count);
        last = first ? std::min<uint32_t>(first + 2, kFinishGeometries) : 0u;
    }
    if (first_capacity) *first_capacity = finish_geometry_capacity(first);
    if (last_capacity) *last_capacity = finish_geometry_capacity(last);
    return GLU_OK;
}

glu_status glu_radix_sort_read_seg_finish(glu_radix_sort sort, uint32_t* attempted, uint32_t* accepted, uint32_t* longest_run,
                                          uint32_t* capacity, uint32_t* runs, uint32_t* tile, uint32_t* split)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    const bool tried = sort->last_seg_finish_attempted && sort->seg_gate.ptr;
    uint32_t longest = 0;
    if (tried) HIP_TRY(hipMemcpy(&longest, sort->seg_gate.ptr, sizeof(longest), hipMemcpyDeviceToHost));
    if (attempted) *attempted = tried ? 1u : 0u;
    if (accepted) *accepted = tried && longest <= sort->last_seg_finish_capacity ? 1u : 0u;
    if (longest_run) *longest_run = longest;
    if (capacity) *capacity = tried ? sort->last_seg_finish_capacity : 0u;
    if (runs) *runs = tried ? sort->last_seg_finish_runs : 0u;
    if (tile) *tile = tried ? sort->last_seg_finish_tile : 0u;
    if (split) *split = tried ? sort->last_seg_finish_split : 0u;
    return GLU_OK;
}

glu_status glu_radix_sort_read_long_runs(glu_radix_sort sort, uint32_t* runs, uint32_t* sub_blocks, uint32_t* pairs)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    uint32_t hdr[3] = {0, 0, 0};
    const bool tried = sort->plan.ptr && sort->last_planned && sort->last_finish_attempted && sort->last_finish_long_ok && sort->long_hdr.ptr;
    if (tried) HIP_TRY(hipMemcpy(hdr, sort->long_hdr.ptr, sizeof(hdr), hipMemcpyDeviceToHost));
    if (runs) *runs = hdr[0];
    if (sub_blocks) *sub_blocks = hdr[1];
    if (pairs) *pairs = hdr[2];
    return GLU_OK;
}

glu_status glu_radix_sort_read_finish(glu_radix_sort sort, uint32_t* attempted, uint32_t* accepted, uint32_t* longest_run,
                                      uint32_t* capacity, uint32_t* top_bit)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    const bool tried = sort->plan.ptr && sort->last_planned && sort->last_finish_attempted; // (no plan yet: nothing was tried)
    PassPlan host;
    memset(&host, 0, sizeof(host));
    if (tried) HIP_TRY(hipMemcpy(&host, sort->plan.ptr, sizeof(host), hipMemcpyDeviceToHost));
    if (attempted) *attempted = tried ? 1u : 0u;
    if (accepted) *accepted = tried && host.finish ? 1u : 0u;
    if (longest_run) *longest_run = tried ? host.finish_longest : 0u;
    if (tried && sort->last_device_top && host.top_bit) sort->last_finish_top = host.top_bit; // (chosen on the device)
    // the tile the device chose; refused: the largest one that was enqueued
    if (capacity) *capacity = !tried ? 0u : host.finish ? finish_geometry_capacity(host.finish) : sort->last_finish_capacity;
    if (top_bit) *top_bit = tried ? sort->last_finish_top : 0u;
    return GLU_OK;
}

namespace
{
// Sums the recorded intervals.  Sorts that tried to end in LDS (radix_lds_finish.hpp) enqueue two sequences of passes, one of
// which returns at once; which one is read from the plan of the LAST sort and assumed for every sort since the previous
// read.  Booked are the passes that did the work: accepted -- the two top-bit passes (count / scan / scatter) and the in-LDS
// pass (finish_ms); refused -- the ordinary passes, plus the count interval of the first top-bit pass (the price of asking).
glu_status read_profile_impl(glu_radix_sort sort, double* count_ms, double* scan_ms, double* scatter_ms, uint64_t* passes,
                             double* finish_ms, uint64_t* finish_passes)
{
    GLU_TRY(enter());
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    double acc[3] = {0, 0, 0}, fin = 0;
    uint64_t live = 0, fin_n = 0;
    const size_t n = sort->slots.size() / 4;
    if (sort->events_used > 0) HIP_TRY(hipEventSynchronize(sort->events[sort->events_used - 1]));
    bool accepted = false;
    if (sort->last_planned && sort->last_finish_attempted && sort->plan.ptr)
    {
        PassPlan host;
        HIP_TRY(hipMemcpy(&host, sort->plan.ptr, sizeof(host), hipMemcpyDeviceToHost));
        accepted = host.finish != 0;
    }
    // the outcome of every attempt of the window, by attempt number (a window of more than 256 sorts: the older ones by the last)
    std::vector<uint32_t> ring(256, 0u);
    const bool have_ring = sort->finish_outcomes.ptr != nullptr;
    if (have_ring) HIP_TRY(hipMemcpy(ring.data(), sort->finish_outcomes.ptr, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const bool last_accepted = accepted;
    bool seg_accepted = false;
    if (sort->last_seg_finish_attempted && sort->seg_gate.ptr)
    {
        uint32_t longest = 0;
        HIP_TRY(hipMemcpy(&longest, sort->seg_gate.ptr, sizeof(longest), hipMemcpyDeviceToHost));
        seg_accepted = longest <= sort->last_seg_finish_capacity;
    }
    for (size_t p = 0; p < n; p++)
    {
        float ms[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; k++) // (light profiling: an interval without both of its events reads as zero)
            if (sort->slots[p * 4 + k] && sort->slots[p * 4 + k + 1])
                HIP_TRY(hipEventElapsedTime(&ms[k], sort->slots[p * 4 + k], sort->slots[p * 4 + k + 1]));
        const uint8_t kind = p < sort->pass_kinds.size() ? sort->pass_kinds[p] : 0;
        const uint32_t seq = p < sort->pass_seqs.size() ? sort->pass_seqs[p] : 0u;
        // (this pass's sort: accepted?  ring entry = attempt number << 3 | tile geometry)
        accepted = last_accepted;
        if (seq && have_ring && (ring[seq & 255u] >> 3) == seq) accepted = (ring[seq & 255u] & 7u) != 0u;
        const bool attempted = seq != 0 || sort->last_finish_attempted;
        // (light profiling records no events around the sequence the host expects not to run: such a pass is not a live pass with
        // zero time, whatever the device then decided)
        const bool timed = sort->slots[p * 4 + 2] && sort->slots[p * 4 + 3];
        if (kind == 2)
        {
            if (accepted || seg_accepted) fin += ms[2], fin_n++;
            continue;
        }
        if (kind == 3 && !seg_accepted)
        {
            acc[0] += ms[0], acc[1] += ms[1]; // its count and scan ran before the device said no
            continue;
        }
        if (kind == 4 && seg_accepted) continue; // the sequence not taken
        if (kind == 1 && !accepted)
        {
            acc[0] += ms[0]; // the leader's count kernel ran before the device said no
            continue;
        }
        if (kind == 0 && accepted && attempted) continue; // the sequence not taken
        if (!timed && sort->profiling == 2) continue;
        for (int k = 0; k < 3; k++) acc[k] += ms[k];
        live++;
    }
    sort->events_used = 0;
    sort->slots.clear();
    sort->pass_kinds.clear();
    sort->pass_seqs.clear();
    if (count_ms) *count_ms = acc[0];
    if (scan_ms) *scan_ms = acc[1];
    if (scatter_ms) *scatter_ms = acc[2];
    if (passes) *passes = live;
    if (finish_ms) *finish_ms = fin;
    if (finish_passes) *finish_passes = fin_n;
    return GLU_OK;
}
}

glu_status glu_radix_sort_read_profile(glu_radix_sort sort, double* count_ms, double* scan_ms, double* scatter_ms,
                                       uint64_t* passes)
{
    return read_profile_impl(sort, count_ms, scan_ms, scatter_ms, passes, nullptr, nullptr);
}

glu_status glu_radix_sort_read_profile_finish(glu_radix_sort sort, double* count_ms, double* scan_ms, double* scatter_ms,
                                              uint64_t* passes, double* finish_ms, uint64_t* finish_passes)
{
    return read_profile_impl(sort, count_ms, scan_ms, scatter_ms, passes, finish_ms, finish_passes);
}

glu_status glu_radix_sort_scratch_placement(glu_radix_sort sort, uint32_t* candidates, double* chosen_ms, double* slowest_ms)
{
    if (!sort) return fail(GLU_ERROR_INVALID_ARGUMENT, "sort is NULL");
    if (candidates) *candidates = sort->tuned_candidates;
    if (chosen_ms) *chosen_ms = sort->tuned_ms;
    if (slowest_ms) *slowest_ms = sort->tuned_worst_ms;
    return GLU_OK;
}

glu_status glu_radix_sort_scratch_size(glu_radix_sort sort, size_t* bytes)
{
    GLU_TRY(enter());
    if (!sort || !bytes) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *bytes = sort->keys.size + sort->vals.size + sort->table.size + sort->plan.size + sort->pair_t2.size + sort->pair_table.size +
             sort->pair_ranges.size + sort->pair_sub.size + sort->seg_desc.size + sort->seg_zero.size + sort->finish_lengths.size +
             sort->finish_starts.size + sort->seg_gate.size + sort->long_image.size + sort->long_hdr.size;
    return GLU_OK;
}

} // extern "C"


#include "glu_dist_impl.hpp"

