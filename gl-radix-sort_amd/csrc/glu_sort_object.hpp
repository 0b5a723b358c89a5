// glu_sort_object.hpp -- the sort object (glu_radix_sort_s), the launch geometries and the arguments of a planned pass: what
// glu_hip.hip (RadixSort's launch sequences) and glu_sort_passes.hpp (the launchers of one counting pass, compiled per key width
// in glu_sort_passes_u32.hip / _u64.hip: 190 kernel instantiations) share.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "glu_host.hpp"
#include "radix_sort_kernels.hpp"

namespace glu_hip
{
namespace host
{
constexpr int kMaxRadix = 256;
constexpr int kMaxBlocksPerCu = 4;

// Launch geometry of one counting pass.  "Large" = one 1024-thread workgroup per CU with the biggest tile LDS
// allows (long per-digit runs, fewest partial 64-byte blocks); "small" = 256-thread workgroups, 4 per CU, for inputs
// that cannot give every CU a large tile.  Measured on MI355X, 2^28 pairs (tools/scatter_bench.hip):
//   8-bit digits: 1024 x 12 + carry 0.95-1.15 ms/pass vs 256 x 16 1.4-1.65 ms;  4-bit digits: 1024 x 12 + carry 0.87 ms vs
//   0.93 ms without the carry (same process), 64-bit keys 1024 x 8 + carry 1.30 vs 1.37 ms.
template<int THREADS_, int KPT_, int BLOCKS_PER_CU_, bool CARRY_>
struct Geometry
{
    static constexpr int THREADS = THREADS_, KPT = KPT_, BLOCKS_PER_CU = BLOCKS_PER_CU_, TILE = THREADS_ * KPT_;
    static constexpr bool CARRY = CARRY_;
};
template<typename KeyT, int BITS, bool LARGE>
struct PairGeometry;
template<> struct PairGeometry<uint32_t, 8, true> : Geometry<1024, 12, 1, true> {};
template<> struct PairGeometry<uint32_t, 4, true> : Geometry<1024, 12, 1, true> {};
// Small geometry of 32-bit keys (tuning builds override these three; tools/small_geometry_sweep.py).  512 x 8 instead of
// round 1's 256 x 16: the same 4096-pair tile ranked, staged and written by twice the lanes -- a pass of a launch-bound sort
// is one tile's latency chain per workgroup: 2^16 pairs 48 -> 43 us, 2^20: 72 -> 66 us, 3 M: 105 -> 97 us (1024 x 4 and
// 2048-pair tiles lose again from 2^20 up; three workgroups per CU beat two and four).
#ifndef GLU_SMALL_THREADS
#define GLU_SMALL_THREADS 512
#define GLU_SMALL_KPT 8
#define GLU_SMALL_BLOCKS_PER_CU 3
#endif
template<> struct PairGeometry<uint32_t, 8, false> : Geometry<GLU_SMALL_THREADS, GLU_SMALL_KPT, GLU_SMALL_BLOCKS_PER_CU, false> {};
template<> struct PairGeometry<uint32_t, 4, false> : Geometry<GLU_SMALL_THREADS, GLU_SMALL_KPT, GLU_SMALL_BLOCKS_PER_CU, false> {};
template<> struct PairGeometry<uint64_t, 8, true> : Geometry<512, 16, 1, true> {};
template<> struct PairGeometry<uint64_t, 4, true> : Geometry<1024, 8, 1, true> {};
template<> struct PairGeometry<uint64_t, 8, false> : Geometry<256, 8, 4, false> {};
template<> struct PairGeometry<uint64_t, 4, false> : Geometry<256, 8, 4, false> {};
// VALS = false (keys-only sorts): the LDS arrays hold keys alone, so the large 32-bit / 8-bit-digit geometry takes 20 keys
// per thread instead of 12 (tile 20480, 127 VGPRs, no spills; scatter 0.66 ms vs 0.72 ms at 12 for 2^28 keys); the
// others keep their tile shape and just drop the value half of every array.
template<typename KeyT, int BITS, bool LARGE, bool VALS = true>
struct GeometryFor : PairGeometry<KeyT, BITS, LARGE> {};
template<> struct GeometryFor<uint32_t, 8, true, false> : Geometry<1024, 20, 1, true> {};

// 128-byte-line scatter (radix_scatter_lines.hpp): 4-byte keys, large inputs, 16-byte aligned arrays.  The carry is
// RADIX x 32 elements of LDS (64 KiB for 8-bit digits of pairs), so the tile is what is left of the 160 KiB: 10 pairs per
// thread (1024 x 10 = 10240), 16 keys per thread for keys-only sorts.
// 8-byte keys: 16-element granules (whole lines of keys, half lines of values), 48 KiB of carry, 8 pairs per thread.
#ifndef GLU_LINES_KPT_U32
#define GLU_LINES_KPT_U32 10 // pairs per thread of the line kernel, 4-byte keys, 8-bit digits (tuning builds override)
#endif
#ifndef GLU_LINES_KPT_U64
#define GLU_LINES_KPT_U64 6 // same, 8-byte keys with values (round 3: 16 KiB of LDS hold back first halves of value lines; 8 before)
#endif
template<typename KeyT, int BITS, bool VALS>
struct LinesGeometry : Geometry<1024, GLU_LINES_KPT_U32, 1, true> {};
// The segmented instantiation of the line kernel (the local sort of the sharded sort, the long runs of a sort that ends in LDS)
// carries the sub-block loop's state on top: with 10 pairs per thread it spills 24 bytes per lane past the 128 registers a
// 1024-thread workgroup has (36 before the kernel's element indices went to 32 bits), with 8 it does not -- and is 5 % SLOWER:
// three segmented passes over 2^27 pairs 1.60-1.66 ms against 1.525-1.533, alternating on one box
// (profiles/r05/seg_scatter_kpt_ab.txt).  The larger tile stays.
#ifndef GLU_LINES_KPT_SEG
#define GLU_LINES_KPT_SEG 10
#endif
struct SegLinesGeometry : Geometry<1024, GLU_LINES_KPT_SEG, 1, true> {};
template<> struct LinesGeometry<uint32_t, 8, false> : Geometry<1024, 16, 1, true> {};
template<> struct LinesGeometry<uint32_t, 4, true> : Geometry<1024, 12, 1, true> {};
template<> struct LinesGeometry<uint32_t, 4, false> : Geometry<1024, 16, 1, true> {};
template<> struct LinesGeometry<uint64_t, 8, true> : Geometry<1024, GLU_LINES_KPT_U64, 1, true> {};
template<> struct LinesGeometry<uint64_t, 8, false> : Geometry<1024, 10, 1, true> {};
#ifndef GLU_LINES_KPT_U64_4BIT
#define GLU_LINES_KPT_U64_4BIT 8 // 8-byte keys with values, 4-bit digits (tuning builds override; 5: no register spills)
#endif
template<> struct LinesGeometry<uint64_t, 4, true> : Geometry<1024, GLU_LINES_KPT_U64_4BIT, 1, true> {};
template<> struct LinesGeometry<uint64_t, 4, false> : Geometry<1024, 10, 1, true> {};
} // namespace host
} // namespace glu_hip

// (glu_radix_sort_s is the C ABI's opaque type: it lives in the global namespace; every translation unit that includes this
// internal header opens the library's namespaces anyway)
using namespace glu_hip;
using namespace glu_hip::host;

struct glu_radix_sort_s
{
    Scratch keys;   // one key scratch array (ping-pong partner of the caller's buffer)
    Scratch vals;
    Scratch table;  // [RADIX][num_blocks] digit counts -> scanned offsets, + RADIX digit totals
    Scratch plan;   // PassPlan of large sorts (which arrays hold the data before each pass, which passes are identities)
    Scratch pair_t2;     // paired passes (radix_pair_passes.hpp): [256][num_blocks][256] 16-bit two-digit counters,
    Scratch pair_table;  // the follower's count table + digit totals,
    Scratch pair_ranges; // the element range of every follower workgroup,
    Scratch pair_sub;    // and (4-bit digits) the leader's table per sub-block: [16][num_blocks * 16]
    Scratch pair_wide;   // the wide rows of a sort that tries to end in LDS: exact counts where 16-bit counters wrapped (kPairWideStride words per block)
    Scratch seg_desc;    // segmented passes (glu_dist's local sort): sub-block descriptors of the pass being enqueued
    Scratch seg_zero;    // and RADIX zero words (the digit totals a segmented scatter adds to its absolute table entries)
    bool last_planned = false; // the last sort on this object ran with a device-side plan (glu_radix_sort_read_plan)
    // pinned host images of the descriptors, a ring: a call fills the next one and enqueues its copy; an image is reused
    // only after the copy enqueued from it has run (its event)
    struct SegStage
    {
        void* host = nullptr;
        size_t size = 0;
        hipEvent_t copied = nullptr;
        bool in_flight = false;
    };
    SegStage seg_stage[16]; // (a sharded sort in rounds makes one segmented sort per round, up to 8: the host never waits for its own sort)
    uint32_t seg_stage_next = 0;
    // a segmented sort that ends in LDS (seg_run_plan): the longest run its first pass found (device word), and what the host knows
    Scratch long_image, long_hdr; // the long runs of a whole-key sort that ends in LDS: their segment descriptors, built on the device
    Scratch long_bits;            // ... and OR / AND of every sub-block's keys (a long run of one key value is left where it is)
    Scratch seg_gate;
    bool seg_finish = true;             // GLU_HIP_SEG_LDS_FINISH=0: always the ordinary segmented passes (tests / tuning)
    bool last_seg_finish_attempted = false;
    uint32_t last_seg_finish_capacity = 0, last_seg_finish_runs = 0, last_seg_finish_tile = 0, last_seg_finish_split = 0;
    // Runs that do not fit the largest tile that still shares a CU (512 x 18 = 9216 pairs) could be taken whole by one workgroup
    // per CU (1024 x 17) or split over 2^k workgroups by key ranges; measured on the runs of the sharded sort at eight ranks
    // (16384 pairs, 2^27 per rank: profiles/r05/force_dist_as_rank_of_8.txt) neither beats the three ordinary passes (1.72 /
    // 1.78 / 2.03 against 1.67 ms), so both are OFF by default and such sorts make no attempt.
    uint32_t seg_split_max = 0;         // GLU_HIP_SEG_SPLIT_MAX: a run is split over at most 2^this workgroups (tests / tuning; up to 3)
    uint32_t seg_split_min = 0;         // GLU_HIP_SEG_SPLIT_MIN: ... and over at least 2^this (tests)
    // the rounds of an in-LDS pass with more than this many bits left to order rank the top 16 .. 23 of them and repair ties
    // (radix_lds_finish.hpp: 64-bit keys, a segmented sort by 32 bits).  GLU_HIP_FINISH_RANK_BITS=N (tuning; 48: all rounds)
    uint32_t finish_rank_bits = 16;
    uint32_t seg_split_geo = 3;         // GLU_HIP_SEG_SPLIT_GEO: the largest tile geometry split runs take (tuning)
    uint32_t seg_max_geo = 4;           // GLU_HIP_SEG_MAX_GEO: the largest tile geometry whole runs take (tests / tuning; up to 5)
    uint32_t digit_bits = 8;
    uint32_t max_blocks = 0;   // GLU_HIP_SORT_BLOCKS: cap on the number of workgroups (tuning)
    uint32_t reserved_cus = 0; // CUs the pass kernels leave free (glu_dist: RCCL kernels run beside them); the grid of a pass
                               // is (CUs - reserved) x workgroups per CU of its geometry
    bool force_small = false;  // GLU_HIP_SORT_SMALL=1: always use the small-tile geometry (tests / tuning)
    bool no_single_block = false; // GLU_HIP_SORT_NO_SINGLE_BLOCK=1: never take the one-workgroup path (tests / tuning)
    bool no_fused_scan = false;   // GLU_HIP_SORT_NO_FUSED_SCAN=1: always launch the row-scan kernel (tests / tuning)
    bool no_plan = false;         // GLU_HIP_SORT_NO_PLAN=1: never skip constant-digit passes (tests / tuning)
    bool tune_scratch = true;     // GLU_HIP_SCRATCH_TUNE=0: take the first allocation of the scratch arrays (see tune_scratch_placement)
    bool tuning = false;          // (inside tune_scratch_placement)
    uint32_t tuned_spacer_mib = 0, tuned_candidates = 0; double tuned_ms = 0, tuned_worst_ms = 0; // what the last tuning saw and chose
    bool no_lines = false;        // GLU_HIP_SORT_NO_LINES=1: never use the 128-byte-line scatter kernel (tests / tuning)
    bool nt_stores = true;        // GLU_HIP_SORT_NT_STORES=0: plain instead of non-temporal line stores in the line scatter (tuning)
    size_t nt_min_bytes = (size_t) 320 << 20; // GLU_HIP_SORT_NT_MIN_BYTES: arrays (keys + values) from this size get the non-temporal stores
    bool no_bit_shortcut = false; // GLU_HIP_SORT_NO_BIT_SHORTCUT=1: passes on key bits that do not vary still count (tests / tuning)
    bool equal_shares = false;    // GLU_HIP_SORT_EQUAL_SHARES=1: line path: equal element shares per workgroup instead of whole tiles (tuning)
    bool pairs = true;            // GLU_HIP_SORT_PAIRS=0: every pass of a large sort counts for itself (tests / tuning)
    uint32_t last_pair_roles[kPlanMaxPasses] = {}; // host-side record of the last planned sort (glu_radix_sort_read_plan)
    size_t pair_min = 0;          // GLU_HIP_SORT_PAIR_MIN=N: element count from which passes are paired (tests / tuning)
    uint32_t pair_unit_div = 16;  // GLU_HIP_SORT_PAIR_UNIT_DIV: a follower counts for itself when a unit of its leader is longer
                                  // than 1 / this of a workgroup's share (0 = never: tests reach the counter-overflow check that way)
    hipEvent_t after_histogram_event = nullptr; // partition passes: recorded once the digit histogram has been copied out
                                                // (after the row scan, before the scatter): glu_dist uses it
    // a sort that ends in LDS (radix_lds_finish.hpp): large whole-key sorts try two top-bit passes + one in-LDS pass
    Scratch finish_lengths;       // [65536] run lengths,
    Scratch finish_starts;        // [65537] run starts
    Scratch finish_crowded;       // the runs the bucket kernel of the in-LDS pass leaves to the ballot rounds (crowded_list_words)
    // (round 6) A sort that tries to end in LDS enqueues two sequences of which the device runs one; the launches of the other
    // return at once, 4.5-5.7 us each -- 22 of them were 105 us behind every accepted attempt (profiles/r05/last_sort_kernels_2p28.txt).
    // They now go to a stream of the object's own that forks off the caller's queue behind the plan kernel and joins it in front
    // of the last kernel: accepted, they return at once UNDER the first top-bit scatter; refused, the kernels left on the
    // caller's queue do.  Forked too: the follower's unit sums (they need the leader's tables only, not its scatter) and the
    // segmented passes over long runs (beside the in-LDS pass, which leaves those runs alone).  Event fork / join: capturable.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_unit = nullptr, ev_fork2 = nullptr, ev_join = nullptr;
    bool fork_behind = true;      // GLU_HIP_SORT_FORK=0: one queue, as in round 5 (tests / tuning)
    bool ensure_side()
    {
        if (side) return true;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) return side = nullptr, false;
        for (hipEvent_t* e : {&ev_fork, &ev_unit, &ev_fork2, &ev_join})
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }
    bool lds_finish = true;       // GLU_HIP_SORT_LDS_FINISH=0: always the four passes of the ordinary sort (tests / tuning)
    bool long_runs = true;        // GLU_HIP_SORT_LONG_RUNS=0: a run longer than the in-LDS pass's tile refuses the whole sort, as in round 4 (tests / tuning)
    size_t finish_min = 0;        // GLU_HIP_SORT_FINISH_MIN=N: element count from which the attempt is made (tests / tuning)
    // A refused attempt costs one read of the keys (4 % of a four-pass sort of 2^28 pairs; nothing when the keys are constant: the
    // ordinary passes then skip on the bits that read collected).  The plan kernel notes each attempt's outcome in a pinned host
    // word (attempt number << 3 | tile geometry, 0 = refused).  EVERY sort asks: what a sort costs does not depend on what the
    // object sorted before (round 4 skipped the next eight attempts after a refusal, and uniform keys behind an all-zero input
    // ran four passes: 4.5 instead of 3.0 ms).  GLU_HIP_SORT_FINISH_BACKOFF=N brings that back: a sort call that finds its LAST
    // attempt refused skips the next N attempts (no synchronisation: an outcome that is not there yet counts as unknown).
    uint32_t* finish_hint = nullptr;
    uint32_t finish_seq = 0, finish_seq_acted_on = 0, finish_wait = 0;
    uint32_t finish_last_geo = 0; // the tile geometry the device chose for the last sort whose outcome is known (0: none yet)
    bool finish_last_refused = false; // the last attempt whose outcome is known was refused (see `side`)
    // The runs of a sort that ends in LDS are the values of the 16 key bits below `top`: the whole key's top 16 by default,
    // the top 16 of the bits that VARIED in this object's last attempt once that is known (keys below 2^28 make 4096 runs of the
    // whole key's top bits and 65536 of bits [12, 28)).  A guess: the plan kernel refuses if a bit from `top` up varies after all.
    bool device_top = true;       // GLU_HIP_SORT_DEVICE_TOP=0: the round-4 rule below (the host guesses from the object's last attempt)
    uint32_t top_floor = 0;       // device_top: the exact top bit of an attempt whose sample missed a varying bit (handed to the next sample)
    bool last_device_top = false; // (glu_radix_sort_read_finish reads the top bit from the device's plan)
    uint32_t finish_top = 0;      // 0: the key's width
    uint32_t last_finish_top = 0; // what the last sort assumed (glu_radix_sort_read_finish)
    uint32_t finish_backoff = 0;  // GLU_HIP_SORT_FINISH_BACKOFF=N (0, the default: every sort attempts)
    bool last_finish_attempted = false; // the last sort enqueued both sequences (glu_radix_sort_read_finish)
    bool last_finish_long_ok = false;   // ... and the segmented passes for runs longer than the tile (glu_radix_sort_read_long_runs)
    uint32_t last_finish_capacity = 0;  // and the longest run its last pass would take
    size_t large_min = 0;         // GLU_HIP_SORT_LARGE_MIN=N: element count from which the large geometry is used (tuning)
    // optional per-kernel timing: 4 marks per pass (before count, after count, after scan, after scatter).  profiling = 1: every
    // mark records an event.  profiling = 2 (LIGHT): only the two marks around the kernel that moves the data of a pass that is
    // expected to run -- the scatter of a pass, the in-LDS pass -- record one; the others are placeholders.  An event between two
    // kernels costs the queue a few microseconds: 28 of them per sort that ends in LDS were 4 % of its time (bench.py's timed
    // region runs light).
    int profiling = 0;
    std::vector<hipEvent_t> events;  // the pool
    size_t events_used = 0;
    std::vector<hipEvent_t> slots;   // one per mark: the event recorded there, or nullptr
    hipEvent_t next_event()
    {
        if (events_used == events.size())
        {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            events.push_back(e);
        }
        return events[events_used++];
    }
    // what the four events of a pass belong to: 0 = a pass of the ordinary sort, 1 = a top-bit pass of a sort that tries to
    // end in LDS, 2 = its in-LDS pass (glu_radix_sort_read_profile books them by which of the two sequences ran); 3 = the
    // counting pass of a SEGMENTED sort that tries to end in LDS, 4 = an ordinary segmented pass enqueued behind such an attempt
    std::vector<uint8_t> pass_kinds;
    // (round 6) which attempt a pass belongs to (finish_seq; 0: none): glu_radix_sort_read_profile books every sort of a window by
    // ITS outcome -- the plan kernel notes them in a ring of 256 (finish_outcomes) -- not by the last sort's
    std::vector<uint32_t> pass_seqs;
    uint32_t cur_seq = 0;
    Scratch finish_outcomes;
    uint8_t cur_kind = 0;
    bool cur_behind = false; // the pass being enqueued belongs to the sequence that is expected NOT to run
    // around_data_kernel: this mark is one of the two around the scatter kernel / the in-LDS pass
    void mark(hipStream_t stream, bool around_data_kernel = false)
    {
        if (!profiling) return;
        if (slots.size() % 4 == 0) pass_kinds.push_back(cur_kind), pass_seqs.push_back(cur_seq);
        hipEvent_t e = nullptr;
        if (profiling == 1 || (around_data_kernel && !cur_behind))
            if ((e = next_event()) != nullptr) (void) hipEventRecord(e, stream);
        slots.push_back(e);
    }
};

namespace glu_hip
{
namespace host
{
// paired passes need the line kernel and the device-side plan, and pay 64 MiB of table traffic and two launches per pair
// of passes whatever the input size: below this many bytes of keys a second read of the keys is cheaper
// (tools/pairs_ladder.py; GLU_HIP_SORT_PAIR_MIN=elements overrides: tests, tuning)
constexpr size_t kPairMinKeyBytes = (size_t) 1 << 28;
// a sort that tries to end in LDS (radix_lds_finish.hpp) pairs its two top-bit passes and pays the same tables, but replaces
// more: it is faster from about 2^24.5 pairs up (tools/finish_midsize_probe.py: 2^25 0.637 -> 0.586 ms, 2^25.5 0.907 -> 0.736)
// (64-bit keys, where it replaces six passes, not two: from about 2^23: 2^24 1.01 -> 0.74 ms, 2^25 1.81 -> 1.07)
// (round 5, with the launches that the long-run passes and the sample added: 32-bit keys still from 2^24.8, 2^25: 0.638 -> 0.578 ms;
// 64-bit keys from 2^23: 0.615 -> 0.474 ms; profiles/r05/finish_midsize_any*.txt, host time from the call to the end of the sort)
// (later in round 5, on a finer ladder: 64-bit keys level at 6.3 M pairs or keys, 3-21 % ahead from 6.6 M .. 7.9 M: profiles/r05/finish_from_u64.txt)
// (32-bit keys with values, once the line stores of sorts this size had become plain ones: level at 27.9 M pairs, 3.5 % ahead at 30.4 M,
// 4.7 % at 33.1 M: profiles/r05/finish_from_u32_pairs.txt -- from 7 * 2^22; keys-only sorts of 32-bit keys reach the line kernel, and with it
// the attempt, at 2^25 keys)
constexpr size_t finish_min_count(size_t key_size) { return key_size == 8 ? (size_t) 3 << 21 : (size_t) 7 << 22; }
constexpr size_t kPlanMinCount = (size_t) 1 << 22; // planned sorts: see PlanArgs below

// CUs the pass kernels of `s` may fill (glu_dist reserves some for RCCL kernels that run beside them)
inline uint32_t usable_cus(const glu_radix_sort_s* s)
{
    return (uint32_t) std::max<int>(1, g_dev.num_cus - (int) s->reserved_cus);
}

// Planned sorts (count >= kPlanMinCount): the kernels pick source / destination from the device-side PassPlan and skip
// the scatter of passes whose digit is constant over the input (radix_sort_kernels.hpp).  `may_skip` is false for the
// passes that must run whatever the data looks like (key encode / decode passes of typed sorts).
struct PlanArgs
{
    PassPlan* plan = nullptr;
    uint32_t pass = 0;
    bool may_skip = false;
    // paired passes (radix_pair_passes.hpp): 1 = leader (its count kernel also builds the two-digit table on the digit
    // [shift2, shift2 + bits2) of the pass after it), 2 = follower (its count table comes from that)
    int pair_role = 0;
    uint32_t shift2 = 0, bits2 = 0;
    uint32_t flags = 0; // kPlanCollectBits / kPlanShortcut for the pass's count kernel
    // the first top-bit pass of a sort that tries to end in LDS (radix_lds_finish.hpp): between its row scan and its scatter
    // the run lengths are summed from the two-digit table and the device decides which sequence of passes runs
    bool behind_attempt = false;  // an ordinary pass enqueued behind such an attempt (runs only if the attempt was refused)
    uint32_t finish_geo_first = 0, finish_geo_last = 0; // tile geometries of the in-LDS pass that are enqueued (0: not such a pass)
    uint32_t finish_first_ordinary = 0, finish_num_ordinary = 0;
    uint32_t finish_seq = 0, finish_top_bit = 0, finish_key_bits = 0;
    bool finish_long_ok = false; // runs longer than the tile go to segmented passes (sort_bits enqueues them behind the in-LDS pass)
    bool fork_side = false;      // behind the plan kernel of this pass the object's side stream forks off (glu_radix_sort_s::side)
    bool unitsum_side = false;   // this follower's unit sums run on the side stream, under its leader's scatter
};

constexpr size_t kSmallResidentPerCU = 4; // workgroups of the small pair geometry (512 x 8) that share a CU (measured: the step in sort time sits at 256 x 4 x 4096 pairs)

// The element count from which a pass runs a large-tile kernel (the 128-byte-line scatter, or the large geometry that arrays
// it cannot take fall back to) instead of the small geometry.  Three rules, all measured on 256 CUs:
//   * 3/2 large tiles per CU: with one to one-and-a-half tiles per workgroup a few workgroups get two tiles and set the kernel
//     time (3.2 M pairs: 145 us large vs 117 us small, 4.2 M: 149 vs 132, 6 M: 162 vs 188);
//   * 4-byte keys with values: not before the small geometry needs a second round of workgroups -- its 4096-element tiles sit
//     four to a CU, and up to that many the whole sort is one tile's latency chain per pass (4.01 M pairs 122 us small against
//     146 us by lines, 4.26 M 158 against 154: profiles/r05/geometry_switch_pairs.txt);
//   * keys-only sorts of 4-byte keys, 8-bit digits: the line kernel's 16 384-key tiles, one workgroup per CU, make sort time a
//     staircase with steps of 4.2 M keys, and the small geometry stays ahead of or level with it up to 2^25 keys (6 .. 30 M
//     keys: 0-15 % by where on a step the size falls; from 37 M the line kernel wins by 7 % and more:
//     profiles/r05/geometry_switch_keys_only.txt).
template<typename KeyT, int BITS>
size_t large_tiles_from(const glu_radix_sort_s* s, bool vals, size_t large_tile)
{
    if (s->large_min) return s->large_min; // GLU_HIP_SORT_LARGE_MIN (tests / tuning)
    size_t from = (size_t) g_dev.num_cus * large_tile * 3 / 2;
    const size_t small_tile = vals ? GeometryFor<KeyT, BITS, false, true>::TILE : GeometryFor<KeyT, BITS, false, false>::TILE;
    if (sizeof(KeyT) == 4 && vals) from = std::max<size_t>(from, (size_t) g_dev.num_cus * kSmallResidentPerCU * small_tile + 1);
    if (sizeof(KeyT) == 4 && !vals && BITS == 8) from = std::max<size_t>(from, (size_t) 1 << 25);
    return from;
}

// does a pass over these arrays run the line kernel?  (launch_pass_sized and the pairing of passes in sort_bits)
template<typename KeyT, int BITS>
bool lines_applicable(const glu_radix_sort_s* s, const void* src_k, const void* src_v, const void* dst_k, const void* dst_v, size_t count)
{
    const bool vals = src_v != nullptr;
    // whole-line stores need 16-byte aligned destinations (hipMalloc gives 256); both pairs of arrays are checked
    // because a planned sort swaps their roles on the device.
    const size_t lines_tile = vals ? LinesGeometry<KeyT, BITS, true>::TILE : LinesGeometry<KeyT, BITS, false>::TILE;
    const bool aligned = (((uintptr_t) src_k | (uintptr_t) src_v | (uintptr_t) dst_k | (uintptr_t) dst_v) & 15u) == 0;
    // (at least one whole tile: the kernel's branch-free prefetch reads tile 0 when it has nothing better to read)
    return aligned && !s->no_lines && !s->force_small && count >= lines_tile && count >= large_tiles_from<KeyT, BITS>(s, vals, lines_tile);
}

// One counting pass (count -> row scan -> scatter, or the leader / follower of a pair of passes), 4- or 8-bit digits, the geometry
// by size: glu_sort_passes.hpp, instantiated for 4-byte keys in glu_sort_passes_u32.hip and for 8-byte keys in _u64.hip.
template<typename KeyT>
glu_status dispatch_pass(glu_radix_sort_s* s, const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v,
                         size_t count, uint32_t shift, uint32_t bits, uint32_t* histogram_out, hipStream_t stream,
                         uint32_t xform = 0, PlanArgs pa = PlanArgs());
extern template glu_status dispatch_pass<uint32_t>(glu_radix_sort_s*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, size_t, uint32_t,
                                                   uint32_t, uint32_t*, hipStream_t, uint32_t, PlanArgs);
extern template glu_status dispatch_pass<uint64_t>(glu_radix_sort_s*, const uint64_t*, const uint32_t*, uint64_t*, uint32_t*, size_t, uint32_t,
                                                   uint32_t, uint32_t*, hipStream_t, uint32_t, PlanArgs);
} // namespace host
} // namespace glu_hip
