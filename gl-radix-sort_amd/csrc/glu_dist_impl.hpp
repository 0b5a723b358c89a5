// glu_dist_impl.hpp -- the sharded (multi-GPU) radix sort behind the glu_dist_* entry points of include/glu_hip.h.
// Included by glu_hip.hip (same translation unit: it uses the library's pass launchers and scratch objects).
//
// The reference is single-device (replicas only); BASELINE.json configs[3] asks for the
// sharded form: rank r holds slice r of the array, one exchange step on the top 8 key bits, then local sorts.
//   1. stable partition of the local slice by the bucket key >> 24 with the sort's own count / scan / scatter kernels
//      (the 256-bin bucket histogram falls out of the row scan, before the scatter runs);
//   2. ncclAllGather of the R x 256 histograms and the copy to the host run on a side stream while the partition's
//      scatter kernel is still running (it leaves eight CUs free, one per XCD, for the RCCL kernel): the host has the plan by the time
//      the partition is done and posts the exchange right behind it, the sort's stream never waits for the host;
//   3. every rank derives the same contiguous bucket -> rank map (a bucket is never split) and its send / receive counts;
//   4. one grouped exchange (ncclGroupStart .. ncclGroupEnd) carries keys and values to and from every peer, receive
//      segments in source-rank order; the part that stays on the rank is a device copy.  Large shards on more than one rank
//      post it in ROUNDS (round 4): every rank's buckets are cut into a few groups, round j is the grouped exchange of
//      group j of every destination, on the side stream, and step 5 runs group by group, each behind its own round;
//   5. local stable sort of what was received: the shard arrives as one message per source rank, each grouped by bucket,
//      and is sorted by its low 24 bits per bucket with three SEGMENTED passes (radix_seg_passes.hpp; the first one reads
//      the messages where they lie, so regrouping them by bucket costs no pass).  Shards that are small, made of very many
//      tiny pieces, or partitioned on a lower byte take the ordinary sort of all 32 bits instead.
// Errors are collective: what a rank finds wrong before a collective (arguments, sizes, allocations) travels with the data of
// that collective -- a status word in the histogram rows of glu_dist_sort_begin, a one-word all-gather in front of the exchange
// of glu_dist_sort_finish -- and every rank returns the failure before anything is sent or received.  What all ranks can
// compute alike (every rank's shard size) is checked by all of them.  A failing RCCL or HIP call after that point is not
// recoverable (the communicator is broken); it is reported by the rank that sees it.
// Concatenating the ranks' outputs in rank order equals the single-device stable sort: equal keys share a bucket, a
// bucket has one owner, the partition and the local sort are stable, and receive segments keep (source rank, source
// index) order.  RCCL is bound with dlopen at first use, so the library has no link-time dependency on it and a process
// that already carries a librccl (PyTorch's) keeps using that one.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace
{
constexpr int kDistBuckets = 256;
constexpr uint32_t kDistTopBits = 8;
// words per rank in the histogram all-gather: the 256 bucket counts, then the slice's element count (low, high word) and the
// rank's status so far (0 = fine): every rank sees every rank's row and takes the same decisions
constexpr int kDistRow = 264;
constexpr int kDistRowCountLo = 256, kDistRowCountHi = 257, kDistRowStatus = 258;
constexpr uint64_t kDistShardLimit = 0xFFFF0000ull; // pairs one device sorts (32-bit indexing)
constexpr size_t kDistSegMinDefault = (size_t) 1 << 24; // shards below this take the ordinary local sort

struct RcclApi
{
    void* lib = nullptr;
    std::string error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {glu_env("GLU_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names)
        {
            if (!name || !*name) continue;
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
            api.error = dlerror();
        }
        if (!api.lib) return;
        auto sym = [&](const char* n) {
            void* p = dlsym(api.lib, n);
            if (!p) api.error = std::string("missing RCCL symbol ") + n;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId)) sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank)) sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy)) sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString)) sym("ncclGetErrorString");
        api.AllGather = (decltype(api.AllGather)) sym("ncclAllGather");
        api.Send = (decltype(api.Send)) sym("ncclSend");
        api.Recv = (decltype(api.Recv)) sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart)) sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd)) sym("ncclGroupEnd");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GetErrorString || !api.AllGather || !api.Send ||
            !api.Recv || !api.GroupStart || !api.GroupEnd)
        {
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return api;
}

glu_status rccl_ready()
{
    RcclApi& r = rccl();
    if (!r.lib) return fail(GLU_ERROR_DEVICE, "RCCL is not available (%s)", r.error.empty() ? "librccl.so not found" : r.error.c_str());
    return GLU_OK;
}

#define NCCL_TRY(expr)                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess) return fail(GLU_ERROR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r_));        \
    } while (0)

// ---- planning (host only; pure functions of the gathered histograms, identical on every rank) ----------------------

// Contiguous, monotone bucket -> rank map balancing the element counts: rank r gets the buckets [cut[r], cut[r + 1]),
// cut[r] = the bucket boundary whose prefix count is nearest to r * N / R (ties to the lower boundary), never before
// cut[r - 1].  A bucket is never split: a single hot bucket bounds the balance.
void dist_plan_buckets(const uint32_t* all_hist, int world, int* owner)
{
    uint64_t prefix[kDistBuckets + 1];
    prefix[0] = 0;
    for (int b = 0; b < kDistBuckets; b++)
    {
        uint64_t t = 0;
        for (int r = 0; r < world; r++) t += all_hist[(size_t) r * kDistBuckets + b];
        prefix[b + 1] = prefix[b] + t;
    }
    const uint64_t n = prefix[kDistBuckets];
    std::vector<int> cuts(world + 1, 0);
    for (int r = 1; r < world; r++)
    {
        // n * r fits 64 bits: n < 2^32 per rank * at most 2^16 ranks
        const uint64_t target = n * (uint64_t) r / (uint64_t) world;
        int b = (int) (std::lower_bound(prefix, prefix + kDistBuckets + 1, target) - prefix); // first boundary >= target
        if (b > kDistBuckets) b = kDistBuckets;
        if (b > 0)
        {
            const uint64_t below = target - prefix[b - 1], above = prefix[b] - target;
            if (below <= above) b -= 1;
        }
        cuts[r] = std::max(b, cuts[r - 1]);
    }
    cuts[world] = kDistBuckets;
    for (int r = 0; r < world; r++)
        for (int b = cuts[r]; b < cuts[r + 1]; b++) owner[b] = r;
}

// true if at most one bucket is non-empty over all ranks: a partition on this digit would send everything to one rank
bool dist_single_bucket(const uint32_t* all_hist, int world)
{
    int used = 0;
    for (int b = 0; b < kDistBuckets && used < 2; b++)
    {
        bool any = false;
        for (int r = 0; r < world && !any; r++) any = all_hist[(size_t) r * kDistBuckets + b] != 0;
        used += any ? 1 : 0;
    }
    return used < 2;
}

// The buckets of every rank cut into `rounds` contiguous GROUPS of about equal element counts (a bucket is never split; groups
// may be empty): cut[q * (rounds + 1) + j] = first bucket of group j of rank q, cut[... + rounds] = one past its last bucket.
// A pure function of what every rank knows after the histogram all-gather, so sender and receiver agree on every message.
void dist_plan_groups(const uint32_t* all_hist, int world, const int* owner, int rounds, int* cut)
{
    for (int q = 0; q < world; q++)
    {
        int g0 = kDistBuckets, g1 = 0;
        for (int b = 0; b < kDistBuckets; b++)
            if (owner[b] == q) g0 = std::min(g0, b), g1 = std::max(g1, b + 1);
        int* c = cut + (size_t) q * (rounds + 1);
        if (g1 <= g0)
        {
            for (int j = 0; j <= rounds; j++) c[j] = 0;
            continue;
        }
        uint64_t total = 0;
        for (int b = g0; b < g1; b++)
            for (int s = 0; s < world; s++) total += all_hist[(size_t) s * kDistBuckets + b];
        c[0] = g0;
        uint64_t before = 0; // elements in buckets [g0, b)
        int b = g0;
        for (int j = 1; j < rounds; j++)
        {
            const uint64_t target = total * (uint64_t) j / (uint64_t) rounds;
            while (b < g1)
            {
                uint64_t in_b = 0;
                for (int s = 0; s < world; s++) in_b += all_hist[(size_t) s * kDistBuckets + b];
                if (before + in_b / 2 > target) break; // the boundary nearer to the target
                before += in_b;
                b++;
            }
            c[j] = b;
        }
        c[rounds] = g1;
    }
}

// send[d] = elements of `rank` whose bucket belongs to rank d; recv[s] = elements of rank s whose bucket belongs to `rank`
void dist_plan_counts(const uint32_t* all_hist, int world, int rank, const int* owner, uint64_t* send, uint64_t* recv)
{
    for (int r = 0; r < world; r++) send[r] = recv[r] = 0;
    for (int b = 0; b < kDistBuckets; b++)
    {
        send[owner[b]] += all_hist[(size_t) rank * kDistBuckets + b];
        if (owner[b] == rank)
            for (int s = 0; s < world; s++) recv[s] += all_hist[(size_t) s * kDistBuckets + b];
    }
}
} // namespace

struct glu_dist_s
{
    int world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    glu_radix_sort_s* sorter = nullptr; // partition pass + local sort (its scratch is sized for the receive side)
    hipStream_t aux = nullptr;          // histogram all-gather + copy to the host, beside the partition's scatter kernel
    hipEvent_t ev_hist = nullptr, ev_plan = nullptr;
    // the exchange in ROUNDS (world > 1, large shards): round j carries the j-th group of every rank's buckets on the side
    // stream while the local sort of group j - 1 runs on the sort's stream (dist_sort_finish)
    static constexpr int kMaxRounds = 8;
    int rounds = 1;                     // glu_dist_set_rounds / GLU_HIP_DIST_ROUNDS (1 .. kMaxRounds); glu_dist_create: 3 from four
                                        // ranks up (at two, half of the data stays on the rank and the one link to the peer
                                        // bounds the exchange far above the local sort: rounds only cost compute there)
    size_t rounds_min = (size_t) 1 << 24; // GLU_HIP_DIST_ROUNDS_MIN: pairs per rank (global count / world) from which rounds are used
    hipEvent_t ev_part = nullptr, ev_round[kMaxRounds] = {};
    uint32_t last_rounds = 1;           // rounds of the last sort's exchange (glu_dist_last_rounds)
    // profiling: one set of events per sort since the last glu_dist_phase_times (start, after partition, after exchange,
    // after local sort on the sort's stream; begin / end of the histogram exchange on the side stream); nothing waits
    // for them before glu_dist_phase_times does
    struct Marks
    {
        hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    };
    std::vector<Marks> marks;
    size_t marks_used = 0;
    Scratch part_k, part_v;             // the local slice grouped by bucket (send side)
    Scratch recv_k, recv_v;             // receive side of glu_dist_sort_ptr (glu_dist_sort_finish takes the caller's)
    Scratch retired_k, retired_v;       // receive arrays that glu_dist_sort_ptr outgrew: alive until the call after (its results alias them)
    Scratch land_k, land_v;             // where the exchange lands when the local sort is segmented (round 5: the sorter's scratch
                                        // arrays are the third pair that a segmented sort that ends in LDS passes through)
    Scratch hist;                       // [kDistRow] local row, [world * kDistRow] gathered rows, [4 + world] status words
    uint32_t* all_hist_host = nullptr;  // pinned: the gathered rows; behind them 8 words of trailer staging, 4 + world of status
    std::vector<uint32_t> hist_dense;   // [world][256] the bucket counts of the gathered rows
    uint64_t shard_limit = kDistShardLimit; // GLU_HIP_DIST_TEST_SHARD_LIMIT lowers it (tests of the collective error path)
    size_t seg_min = kDistSegMinDefault;    // GLU_HIP_DIST_SEG_MIN: shard size from which the local sort is segmented
    bool seg_enabled = true;                // GLU_HIP_DIST_SEG=0: always the ordinary local sort
    bool seg_forced = false;                // GLU_HIP_DIST_SEG=2: segmented whenever the shard has 2^16 pairs, however fragmented (tests)
    int test_fail_begin = -1, test_fail_finish = -1; // GLU_HIP_DIST_TEST_FAIL=begin:<rank> / finish:<rank>: that rank reports a failure
    // GLU_HIP_DIST_TEST_FAULT: a stream dependency is left out on purpose (negative controls of tests/test_gpu_dist.py).
    // 1 = no_hist_wait: the histogram all-gather is not ordered behind the count + scan kernels; 2 = local_sort_unordered: the
    // local sort runs on the side stream, not behind the exchange (right with a transport that completes inside the call)
    int test_fault = 0;
    uint32_t last_local_sort = 0;           // 1 = the last sort's local sort was segmented, 0 = ordinary (glu_dist_last_local_sort)
    std::vector<int> owner;
    std::vector<uint64_t> send_counts, recv_counts;
    size_t local_count = 0;
    uint64_t recv_total = 0;
    uint32_t partition_shift = 24;      // the key byte the last sort was partitioned on
    bool repartition_at_world_1 = false; // GLU_HIP_DIST_TEST_REPARTITION=1: take the lower-byte fallback with one rank too (tests)
    bool began = false;
    bool profiling = false;
    int reserved_cus = 0;               // glu_dist_set_reserved_cus
};

namespace
{
// the event set of the sort that is being enqueued (profiling on), or nullptr
glu_dist_s::Marks* dist_marks(glu_dist_s* d, bool begin_new)
{
    if (!d->profiling) return nullptr;
    if (begin_new)
    {
        if (d->marks_used == d->marks.size())
        {
            glu_dist_s::Marks m;
            for (hipEvent_t& e : m.e)
                if (hipEventCreate(&e) != hipSuccess) return nullptr;
            d->marks.push_back(m);
        }
        d->marks_used++;
    }
    return d->marks_used ? &d->marks[d->marks_used - 1] : nullptr;
}
} // namespace

extern "C" {

glu_status glu_dist_available(void)
{
    return rccl_ready(); // dlopen + the nine symbols, nothing else: no device call, no bootstrap socket, no thread
}

glu_status glu_dist_unique_id(void* id_out, size_t id_bytes)
{
    GLU_TRY(enter());
    GLU_TRY(rccl_ready());
    if (!id_out || id_bytes < sizeof(ncclUniqueId))
        return fail(GLU_ERROR_INVALID_ARGUMENT, "id_out must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return GLU_OK;
}

glu_status glu_dist_create(const void* unique_id, size_t id_bytes, int world_size, int rank, glu_dist* out)
{
    GLU_TRY(enter());
    GLU_TRY(rccl_ready());
    if (!out) return fail(GLU_ERROR_INVALID_ARGUMENT, "out is NULL");
    if (!unique_id || id_bytes < sizeof(ncclUniqueId)) return fail(GLU_ERROR_INVALID_ARGUMENT, "unique_id must hold %zu bytes", sizeof(ncclUniqueId));
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(GLU_ERROR_INVALID_ARGUMENT, "bad rank %d of %d", rank, world_size);
    glu_dist_s* d = new (std::nothrow) glu_dist_s();
    if (!d) return fail(GLU_ERROR_OUT_OF_MEMORY, "out of host memory");
    d->world = world_size;
    d->rank = rank;
    d->owner.assign(kDistBuckets, 0);
    if (const char* e = glu_env("GLU_HIP_DIST_TEST_REPARTITION")) d->repartition_at_world_1 = atoi(e) != 0;
    if (const char* e = glu_env("GLU_HIP_DIST_TEST_SHARD_LIMIT"))
        if (atoll(e) > 0) d->shard_limit = (uint64_t) atoll(e);
    if (const char* e = glu_env("GLU_HIP_DIST_SEG_MIN")) d->seg_min = (size_t) atoll(e);
    if (const char* e = glu_env("GLU_HIP_DIST_SEG")) d->seg_enabled = atoi(e) != 0, d->seg_forced = atoi(e) == 2;
    d->rounds = world_size >= 4 ? 3 : 1;
    if (const char* e = glu_env("GLU_HIP_DIST_ROUNDS"))
        if (atoi(e) >= 1 && atoi(e) <= glu_dist_s::kMaxRounds) d->rounds = atoi(e);
    if (const char* e = glu_env("GLU_HIP_DIST_ROUNDS_MIN")) d->rounds_min = (size_t) atoll(e);
    if (const char* e = glu_env("GLU_HIP_DIST_TEST_FAULT"))
        d->test_fault = strcmp(e, "no_hist_wait") == 0 ? 1 : (strcmp(e, "local_sort_unordered") == 0 ? 2 : 0);
    if (const char* e = glu_env("GLU_HIP_DIST_TEST_FAIL"))
    {
        if (strncmp(e, "begin:", 6) == 0) d->test_fail_begin = atoi(e + 6);
        if (strncmp(e, "finish:", 7) == 0) d->test_fail_finish = atoi(e + 7);
    }
    d->hist_dense.assign((size_t) world_size * kDistBuckets, 0);
    d->send_counts.assign(world_size, 0);
    d->recv_counts.assign(world_size, 0);
    auto cleanup = [&](glu_status st) {
        glu_dist_destroy(d);
        return st;
    };
    if (glu_status st = glu_radix_sort_create(&d->sorter); st != GLU_OK) return cleanup(st);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    if (ncclResult_t r = rccl().CommInitRank(&d->comm, world_size, id, rank); r != ncclSuccess)
    {
        d->comm = nullptr;
        return cleanup(fail(GLU_ERROR_DEVICE, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)));
    }
    hipError_t e = hipStreamCreateWithFlags(&d->aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_hist, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_plan, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_part, hipEventDisableTiming);
    for (int j = 0; j < glu_dist_s::kMaxRounds && e == hipSuccess; j++) e = hipEventCreateWithFlags(&d->ev_round[j], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**) &d->all_hist_host, ((size_t) world_size * kDistRow + 8 + 4 + world_size) * sizeof(uint32_t));
    if (e != hipSuccess) return cleanup(fail(GLU_ERROR_DEVICE, "glu_dist_create: %s", hipGetErrorString(e)));
    if (glu_status st = d->hist.reserve(((size_t) (world_size + 1) * kDistRow + 4 + world_size) * sizeof(uint32_t)); st != GLU_OK) return cleanup(st);
    *out = d;
    return GLU_OK;
}

glu_status glu_dist_destroy(glu_dist d)
{
    if (!d) return GLU_OK;
    GLU_TRY(enter());
    (void) hipDeviceSynchronize();
    if (d->comm) (void) rccl().CommDestroy(d->comm);
    if (d->sorter) (void) glu_radix_sort_destroy(d->sorter);
    if (d->aux) (void) hipStreamDestroy(d->aux);
    if (d->ev_hist) (void) hipEventDestroy(d->ev_hist);
    if (d->ev_plan) (void) hipEventDestroy(d->ev_plan);
    if (d->ev_part) (void) hipEventDestroy(d->ev_part);
    for (hipEvent_t e : d->ev_round)
        if (e) (void) hipEventDestroy(e);
    for (glu_dist_s::Marks& m : d->marks)
        for (hipEvent_t e : m.e)
            if (e) (void) hipEventDestroy(e);
    if (d->all_hist_host) (void) hipHostFree(d->all_hist_host);
    d->part_k.release();
    d->part_v.release();
    d->recv_k.release();
    d->recv_v.release();
    d->land_k.release();
    d->land_v.release();
    d->retired_k.release();
    d->retired_v.release();
    d->hist.release();
    delete d;
    return GLU_OK;
}

glu_status glu_dist_partition_shift(glu_dist d, uint32_t* shift)
{
    if (!d || !shift) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *shift = d->partition_shift;
    return GLU_OK;
}

glu_status glu_dist_world(glu_dist d, int* world_size, int* rank)
{
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (world_size) *world_size = d->world;
    if (rank) *rank = d->rank;
    return GLU_OK;
}

glu_status glu_dist_local_sorter(glu_dist d, glu_radix_sort* out)
{
    if (!d || !out) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *out = d->sorter; // owned by `d`: for glu_radix_sort_set_digit_bits / set_profiling / read_profile, not for destroy
    return GLU_OK;
}

glu_status glu_dist_prepare(glu_dist d, size_t local_count, size_t recv_capacity)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    // the local sorter first: partition pass: table only; local sort: scratch for the receive side (the landing arrays of the
    // exchange), placed by measurement
    GLU_TRY(sort_prepare(d->sorter, std::max(local_count, recv_capacity), sizeof(uint32_t), true, /*may_place=*/true));
    // then the send-side pair (the partitioned slice) and the receive-side pair of glu_dist_sort_ptr, each placed by
    // measurement against the sorter: every scatter pass of a rank's sort but the first one's source then runs between pairs
    // of arrays that were chosen, not drawn (place_pair_by_measurement; plain allocations for small arrays)
    GLU_TRY(place_pair_by_measurement(d->sorter, std::max<size_t>(local_count, 1), d->part_k, d->part_v));
    if (recv_capacity) GLU_TRY(place_pair_by_measurement(d->sorter, recv_capacity, d->recv_k, d->recv_v));
    // (the landing pair only if a shard of this capacity, 256 / world buckets of equal size, would try to end in LDS)
    {
        uint32_t geo = 0, split_log2 = 0;
        const uint64_t buckets = std::max<uint64_t>(1, (uint64_t) kDistBuckets / (uint64_t) std::max(d->world, 1));
        if (recv_capacity && seg_finish_choice_for(d->sorter, 3u, 24u, buckets * 256u, recv_capacity / buckets, geo, split_log2))
            GLU_TRY(place_pair_by_measurement(d->sorter, recv_capacity, d->land_k, d->land_v));
    }
    // segmented local sort: one table row per sub-block (at most pieces + workgroups: world x buckets owned + CUs) and the
    // descriptor image of its two pass shapes
    const size_t rows = (size_t) kDistBuckets * (size_t) std::min(d->world, 16) + (size_t) g_dev.num_cus;
    GLU_TRY(d->sorter->table.reserve(rows * kDistBuckets * sizeof(uint32_t)));
    GLU_TRY(d->sorter->seg_desc.reserve(std::max<size_t>((size_t) 1 << 16, rows * 6 * sizeof(uint32_t))));
    return GLU_OK;
}

glu_status glu_dist_set_reserved_cus(glu_dist d, int cus)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (cus < 0 || cus * 4 >= g_dev.num_cus) return fail(GLU_ERROR_INVALID_ARGUMENT, "cannot reserve %d of %d CUs", cus, g_dev.num_cus);
    d->reserved_cus = cus;
    d->sorter->reserved_cus = (uint32_t) cus; // every pass: (CUs - reserved) x workgroups per CU of its geometry
    return GLU_OK;
}

glu_status glu_dist_set_profiling(glu_dist d, int enable)
{
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    d->profiling = enable != 0;
    return GLU_OK;
}

glu_status glu_dist_phase_times(glu_dist d, double* ms4, uint64_t* sorts)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    double sum[4] = {0, 0, 0, 0};
    uint64_t n = 0;
    for (size_t i = 0; i < d->marks_used; i++)
    {
        hipEvent_t* e = d->marks[i].e;
        if (hipEventSynchronize(e[3]) != hipSuccess || hipEventSynchronize(e[5]) != hipSuccess) continue;
        float a, b, c, f;
        if (hipEventElapsedTime(&a, e[0], e[1]) != hipSuccess || hipEventElapsedTime(&b, e[4], e[5]) != hipSuccess ||
            hipEventElapsedTime(&c, e[1], e[2]) != hipSuccess || hipEventElapsedTime(&f, e[2], e[3]) != hipSuccess)
            continue;
        sum[0] += a, sum[1] += b, sum[2] += c, sum[3] += f;
        n++;
    }
    d->marks_used = 0;
    for (int i = 0; i < 4; i++)
        if (ms4) ms4[i] = n ? sum[i] / (double) n : 0.0;
    if (sorts) *sorts = n;
    return GLU_OK;
}

glu_status glu_dist_plan_buckets(const uint32_t* all_hist, int world_size, int* bucket_owner)
{
    if (!all_hist || !bucket_owner || world_size < 1) return fail(GLU_ERROR_INVALID_ARGUMENT, "bad arguments");
    dist_plan_buckets(all_hist, world_size, bucket_owner);
    return GLU_OK;
}

glu_status glu_dist_plan_counts(const uint32_t* all_hist, int world_size, int rank, const int* bucket_owner,
                                uint64_t* send_counts, uint64_t* recv_counts)
{
    if (!all_hist || !bucket_owner || !send_counts || !recv_counts || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "bad arguments");
    for (int b = 0; b < kDistBuckets; b++)
        if (bucket_owner[b] < 0 || bucket_owner[b] >= world_size) return fail(GLU_ERROR_INVALID_ARGUMENT, "bucket %d has owner %d", b, bucket_owner[b]);
    dist_plan_counts(all_hist, world_size, rank, bucket_owner, send_counts, recv_counts);
    return GLU_OK;
}

glu_status glu_dist_plan_groups(const uint32_t* all_hist, int world_size, const int* bucket_owner, int rounds, int* group_cut)
{
    if (!all_hist || !bucket_owner || !group_cut || world_size < 1 || rounds < 1 || rounds > glu_dist_s::kMaxRounds)
        return fail(GLU_ERROR_INVALID_ARGUMENT, "bad arguments");
    for (int b = 0; b < kDistBuckets; b++)
        if (bucket_owner[b] < 0 || bucket_owner[b] >= world_size) return fail(GLU_ERROR_INVALID_ARGUMENT, "bucket %d has owner %d", b, bucket_owner[b]);
    dist_plan_groups(all_hist, world_size, bucket_owner, rounds, group_cut);
    return GLU_OK;
}

// Steps 1-3: partition the local slice, exchange the histograms, plan.  The partition is enqueued on `stream`; the call
// returns when the host has the plan (it waited for the histogram exchange on the side stream, not for the partition's
// scatter kernel).  *recv_count = number of pairs this rank will receive: the caller sizes its receive arrays with it.
// Collective: what this rank finds wrong with its own arguments or allocations travels in its histogram row, and every
// rank returns a failure (none is left waiting in a collective).
glu_status glu_dist_sort_begin(glu_dist d, const uint32_t* keys, const uint32_t* vals, size_t local_count, void* stream,
                               size_t* recv_count)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    hipStream_t st = pick_stream(stream);
    d->began = false;
    // rank-local findings: remembered, not returned -- the rank still takes part in the histogram exchange
    glu_status local = GLU_OK;
    std::string local_message;
    auto note = [&](glu_status status) {
        if (local == GLU_OK && status != GLU_OK)
        {
            local = status;
            local_message = g_last_error;
        }
    };
    if (local_count > 0 && (!keys || !vals)) note(fail(GLU_ERROR_INVALID_ARGUMENT, "NULL array"));
    if (local_count > kDistShardLimit) note(fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu does not fit 32-bit indexing", local_count));
    note(d->part_k.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    note(d->part_v.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    {
        uint64_t cap = (uint64_t) g_dev.num_cus * kMaxBlocksPerCu;
        note(d->sorter->table.reserve((kMaxRadix * cap + kMaxRadix) * sizeof(uint32_t)));
    }
    if (d->test_fail_begin == d->rank) note(fail(GLU_ERROR_OUT_OF_MEMORY, "injected failure of rank %d (GLU_HIP_DIST_TEST_FAIL)", d->rank));
    glu_dist_s::Marks* marks = dist_marks(d, true);
    uint32_t* hist = (uint32_t*) d->hist.ptr;
    uint32_t* all_hist = hist + kDistRow;
    uint32_t* trailer_host = d->all_hist_host + (size_t) d->world * kDistRow;
    if (marks) HIP_TRY(hipEventRecord(marks->e[0], st));

    // 1 + 2, on the top key byte first.  If every key of every rank shares that byte (24-bit keys, small integers ...) all
    // the data would land on one rank: the partition is repeated on the next lower byte, down to the lowest.  The bytes
    // above the chosen one are constant over the whole input, so bucket order is still key order and a rank's shard is a
    // contiguous key range.  (The first attempt was then an identity copy of the slice: 0.6 ms at 2^27, paid only by such
    // inputs.)  Every rank sees the same gathered histograms and takes the same decision.
    for (uint32_t shift = 32 - kDistTopBits;; shift -= kDistTopBits)
    {
        // the row's trailer (the host's copy of the previous one has been consumed: the call waited for its all-gather)
        trailer_host[0] = (uint32_t) ((uint64_t) local_count & 0xFFFFFFFFu);
        trailer_host[1] = (uint32_t) ((uint64_t) local_count >> 32);
        trailer_host[2] = (uint32_t) local;
        trailer_host[3] = 0;
        HIP_TRY(hipMemcpyAsync(hist + kDistRowCountLo, trailer_host, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        // 1. stable partition by the bucket (key >> shift) & 255; the histogram is ready (and ev_hist recorded) after the
        //    row scan, before the scatter
        if (local_count == 0 || local != GLU_OK)
        {
            HIP_TRY(hipMemsetAsync(hist, 0, kDistBuckets * sizeof(uint32_t), st));
            HIP_TRY(hipEventRecord(d->ev_hist, st));
        }
        else
        {
            const uint32_t saved_reserved = d->sorter->reserved_cus;
            // The histogram exchange of step 2 needs a CU beside the scatter, and workgroups are dealt round-robin to the 8
            // XCDs: with 254 workgroups six XCDs are full and a one-workgroup kernel bound for one of them waits for the
            // scatter to end (measured: 0.40 ms for the exchange with 2 CUs left, 0.016 ms with 8 = one per XCD).
            const int leave = std::max(d->reserved_cus, 8);
            if (g_dev.num_cus > 4 * leave) d->sorter->reserved_cus = (uint32_t) leave;
            d->sorter->after_histogram_event = d->ev_hist;
            glu_status ps = dispatch_pass<uint32_t>(d->sorter, keys, vals, (uint32_t*) d->part_k.ptr, (uint32_t*) d->part_v.ptr,
                                                    local_count, shift, kDistTopBits, hist, st);
            d->sorter->after_histogram_event = nullptr;
            d->sorter->reserved_cus = saved_reserved;
            GLU_TRY(ps); // (a launch failure: not recoverable, see the head of this file)
        }
        if (marks && shift == 32 - kDistTopBits) HIP_TRY(hipEventRecord(marks->e[1], st));

        // 2. every rank learns every rank's row (R x 264 words): side stream, beside the scatter kernel
        // (test_fault 1 = fault injection for tests/test_gpu_dist.py::test_async_transport_catches_a_missing_stream_dependency:
        // the all-gather is then NOT ordered behind the histogram -- what the asynchronous test double must catch)
        if (d->test_fault != 1) HIP_TRY(hipStreamWaitEvent(d->aux, d->ev_hist, 0));
        if (marks && shift == 32 - kDistTopBits) HIP_TRY(hipEventRecord(marks->e[4], d->aux));
        NCCL_TRY(rccl().AllGather(hist, all_hist, kDistRow, ncclUint32, d->comm, d->aux));
        HIP_TRY(hipMemcpyAsync(d->all_hist_host, all_hist, (size_t) d->world * kDistRow * sizeof(uint32_t), hipMemcpyDeviceToHost, d->aux));
        if (marks) HIP_TRY(hipEventRecord(marks->e[5], d->aux));
        HIP_TRY(hipEventRecord(d->ev_plan, d->aux));
        HIP_TRY(hipEventSynchronize(d->ev_plan));

        // every rank reads every row: the same verdict everywhere
        for (int r = 0; r < d->world; r++)
        {
            const uint32_t* row = d->all_hist_host + (size_t) r * kDistRow;
            if (row[kDistRowStatus] != 0)
            {
                if (r == d->rank)
                {
                    g_last_error = local_message;
                    return local;
                }
                return fail((glu_status) row[kDistRowStatus], "rank %d reported a failure in glu_dist_sort_begin (status %u): no rank sorts", r,
                            row[kDistRowStatus]);
            }
        }
        for (int r = 0; r < d->world; r++)
        {
            const uint32_t* row = d->all_hist_host + (size_t) r * kDistRow;
            uint64_t sum = 0;
            for (int b = 0; b < kDistBuckets; b++) sum += row[b];
            const uint64_t count_r = (uint64_t) row[kDistRowCountLo] | ((uint64_t) row[kDistRowCountHi] << 32);
            if (sum != count_r)
                return fail(GLU_ERROR_INVALID_STATE, "rank %d: bucket histogram sums to %llu, its slice holds %llu", r, (unsigned long long) sum,
                            (unsigned long long) count_r);
            memcpy(&d->hist_dense[(size_t) r * kDistBuckets], row, kDistBuckets * sizeof(uint32_t));
        }
        d->partition_shift = shift;
        if (shift == 0 || (d->world == 1 && !d->repartition_at_world_1) || !dist_single_bucket(d->hist_dense.data(), d->world)) break;
        if (marks) HIP_TRY(hipEventRecord(marks->e[1], st)); // the repeated partition counts as partition time
    }

    // 3. identical plan on every rank; every rank checks every rank's shard size
    dist_plan_buckets(d->hist_dense.data(), d->world, d->owner.data());
    for (int r = 0; r < d->world; r++)
    {
        uint64_t shard = 0;
        for (int b = 0; b < kDistBuckets; b++)
            if (d->owner[b] == r)
                for (int q = 0; q < d->world; q++) shard += d->hist_dense[(size_t) q * kDistBuckets + b];
        if (shard > d->shard_limit)
            return fail(GLU_ERROR_INVALID_ARGUMENT,
                        "rank %d would receive %llu pairs (buckets are never split: a hot bucket bounds the balance), more than one device "
                        "sorts (%llu): no rank sorts",
                        r, (unsigned long long) shard, (unsigned long long) d->shard_limit);
    }
    dist_plan_counts(d->hist_dense.data(), d->world, d->rank, d->owner.data(), d->send_counts.data(), d->recv_counts.data());
    uint64_t total = 0;
    for (int r = 0; r < d->world; r++) total += d->recv_counts[r];
    d->local_count = local_count;
    d->recv_total = total;
    d->began = true;
    if (recv_count) *recv_count = (size_t) total;
    return GLU_OK;
}

} // extern "C"

namespace
{
// The pieces of the shard as it arrives (source-major: the message of source s holds this rank's buckets in ascending
// order), as segments = buckets, in the order (source, bucket): a bucket's pieces are in source order, which is the
// stable order of its elements.
void dist_shard_pieces(const glu_dist_s* d, std::vector<SegPiece>& pieces, uint32_t& nseg)
{
    int g0 = kDistBuckets, g1 = 0;
    for (int b = 0; b < kDistBuckets; b++)
        if (d->owner[b] == d->rank) g0 = std::min(g0, b), g1 = std::max(g1, b + 1);
    nseg = g1 > g0 ? (uint32_t) (g1 - g0) : 0u;
    pieces.clear();
    uint64_t at = 0;
    for (int s = 0; s < d->world; s++)
        for (int b = g0; b < g1; b++)
        {
            const uint64_t len = d->hist_dense[(size_t) s * kDistBuckets + b];
            if (len) pieces.push_back(SegPiece{at, len, (uint32_t) (b - g0)});
            at += len;
        }
}

// Steps 4-5.  `pre_status`: what the caller (glu_dist_sort_ptr) found wrong on this rank before the call.
glu_status dist_sort_finish(glu_dist_s* d, uint32_t* recv_keys, uint32_t* recv_vals, size_t capacity, hipStream_t st, glu_status pre_status)
{
    if (!d->began) return fail(GLU_ERROR_INVALID_STATE, "glu_dist_sort_finish without glu_dist_sort_begin");
    d->began = false;
    const size_t n_recv = (size_t) d->recv_total;
    // rank-local findings first, none returned yet: the ranks agree on them before anything is sent
    glu_status local = pre_status;
    std::string local_message = g_last_error;
    auto note = [&](glu_status status) {
        if (local == GLU_OK && status != GLU_OK)
        {
            local = status;
            local_message = g_last_error;
        }
    };
    if (n_recv > capacity) note(fail(GLU_ERROR_INVALID_ARGUMENT, "receive arrays hold %zu pairs, %zu arrive", capacity, n_recv));
    if (n_recv > 0 && (!recv_keys || !recv_vals)) note(fail(GLU_ERROR_INVALID_ARGUMENT, "NULL receive array"));
    if (d->test_fail_finish == d->rank) note(fail(GLU_ERROR_OUT_OF_MEMORY, "injected failure of rank %d (GLU_HIP_DIST_TEST_FAIL)", d->rank));
    if (local == GLU_OK && n_recv > 1) note(sort_prepare(d->sorter, n_recv, sizeof(uint32_t), true));

    // the local sort: three segmented passes on the low 24 bits (the shard arrives grouped by source rank, then bucket),
    // unless the shard is small, was partitioned on a lower byte, or falls into so many tiny pieces that a workgroup would
    // spend its time on sub-block prologues (about two tiles' time each, whatever the size)
    SegPlan plan;
    bool segmented = false;
    if (local == GLU_OK && d->seg_enabled && d->partition_shift == 32 - kDistTopBits &&
        n_recv >= std::max(d->seg_forced ? (size_t) 0 : d->seg_min, kSegMinCount) &&
        d->sorter->digit_bits == 8 && !d->sorter->no_lines && (((uintptr_t) recv_keys | (uintptr_t) recv_vals) & 15u) == 0)
    {
        std::vector<SegPiece> pieces;
        uint32_t nseg = 0;
        dist_shard_pieces(d, pieces, nseg);
        seg_make_plan(d->sorter, std::move(pieces), nseg, n_recv, 32 - kDistTopBits, true, plan);
        const uint64_t tiles_per_wg = n_recv / LinesGeometry<uint32_t, 8, true>::TILE / usable_cus(d->sorter);
        segmented = !plan.by_copies && (d->seg_forced || plan.max_subs_per_wg() <= 2 + tiles_per_wg / 6);
    }
    // A segmented sort that will try to end in LDS needs a third pair of arrays: the exchange lands in the object's landing pair,
    // the sorter's scratch takes the counting pass, the caller's arrays the result.  A sort that will not try (eight ranks by
    // default: runs of 16384 pairs fit no tile that shares a CU) lands in the sorter's scratch, as in round 4 -- no third pair
    // of shard size is allocated or placed for it (1 GiB per rank at 2^27 pairs).
    bool own_landing = false;
    if (segmented)
    {
        uint32_t geo = 0, split_log2 = 0;
        own_landing = seg_finish_choice(d->sorter, plan, geo, split_log2);
        if (own_landing)
        {
            // (grown here when glu_dist_prepare did not size them: an allocation, agreed on like every other)
            note(d->land_k.reserve(n_recv * sizeof(uint32_t)));
            note(d->land_v.reserve(n_recv * sizeof(uint32_t)));
        }
        else
            note(sort_prepare(d->sorter, n_recv, sizeof(uint32_t), true));
        if (local != GLU_OK) segmented = false;
    }
    d->last_local_sort = segmented ? 1u : 0u;

    if (d->world > 1)
    {
        // one word per rank: is everybody ready to exchange?  (side stream; the partition may still be running on `st`)
        uint32_t* status_dev = (uint32_t*) d->hist.ptr + (size_t) (d->world + 1) * kDistRow;
        uint32_t* status_host = d->all_hist_host + (size_t) d->world * kDistRow + 8;
        status_host[0] = (uint32_t) local;
        HIP_TRY(hipMemcpyAsync(status_dev, status_host, sizeof(uint32_t), hipMemcpyHostToDevice, d->aux));
        NCCL_TRY(rccl().AllGather(status_dev, status_dev + 4, 1, ncclUint32, d->comm, d->aux));
        HIP_TRY(hipMemcpyAsync(status_host + 4, status_dev + 4, (size_t) d->world * sizeof(uint32_t), hipMemcpyDeviceToHost, d->aux));
        HIP_TRY(hipEventRecord(d->ev_plan, d->aux));
        HIP_TRY(hipEventSynchronize(d->ev_plan));
        for (int r = 0; r < d->world; r++)
            if (status_host[4 + r] != 0 && r != d->rank && local == GLU_OK)
                return fail((glu_status) status_host[4 + r], "rank %d reported a failure in glu_dist_sort_finish (status %u): nothing was exchanged", r,
                            status_host[4 + r]);
    }
    if (local != GLU_OK)
    {
        g_last_error = local_message;
        return local;
    }

    const uint32_t* part_k = (const uint32_t*) d->part_k.ptr;
    const uint32_t* part_v = (const uint32_t*) d->part_v.ptr;
    // where the exchange delivers: the segmented sort reads the shard from the object's landing arrays and leaves its
    // result in the caller's (through the sorter's scratch arrays when it ends in LDS); the ordinary sort works in place in the
    // caller's
    uint32_t* land_k = !segmented ? recv_keys : own_landing ? (uint32_t*) d->land_k.ptr : (uint32_t*) d->sorter->keys.ptr;
    uint32_t* land_v = !segmented ? recv_vals : own_landing ? (uint32_t*) d->land_v.ptr : (uint32_t*) d->sorter->vals.ptr;

    // 4. the exchange.  One grouped exchange (ncclSend / ncclRecv to and from every peer between ncclGroupStart / End) per
    // ROUND.  With one round a rank's message to a peer is all of the peer's buckets and everything runs on `st`: partition
    // -> exchange -> local sort in series.  With several rounds (world > 1, large shards: `rounds` is a function of what
    // every rank knows, so all ranks post the same sequence) round j carries GROUP j of every destination's buckets
    // (dist_plan_groups) on the side stream, behind the partition, and the local sort of group j runs on `st` behind round j
    // only: the exchange of the later groups travels while the earlier ones are sorted.  A group of buckets is a contiguous
    // key range of the shard, sorted on its own by the same segmented passes; what lands is group-major, then source rank,
    // then bucket (one sort at a time at eight GPUs: partition + one round + max(rest of the exchange, sorts) instead of
    // partition + exchange + sort).  A rank whose local sort is the ordinary one posts the same rounds
    // into the usual source-major layout and sorts when the last round has landed.
    const int R = d->world;
    uint64_t per_rank = 0;
    for (int q = 0; q < R; q++)
        for (int b = 0; b < kDistBuckets; b++) per_rank += d->hist_dense[(size_t) q * kDistBuckets + b];
    per_rank /= (uint64_t) R;
    const int rounds = ((R > 1 || d->repartition_at_world_1) && d->partition_shift == 32 - kDistTopBits && per_rank >= d->rounds_min) ? d->rounds : 1;
    // (repartition_at_world_1, GLU_HIP_DIST_TEST_REPARTITION=1: a single rank takes the multi-rank code paths too, for tests
    // and for measuring what the rounds cost a rank without a fabric: bench.py --force-dist)
    d->last_rounds = (uint32_t) rounds;
    std::vector<int> cut((size_t) R * (rounds + 1));
    dist_plan_groups(d->hist_dense.data(), R, d->owner.data(), rounds, cut.data());
    // (a rank's partitioned slice is bucket-major: its elements of the buckets [b0, b1) start count_in(rank, 0, b0) in)
    auto hist_of = [&](int q, int b) { return (uint64_t) d->hist_dense[(size_t) q * kDistBuckets + b]; };
    auto count_in = [&](int q, int b0, int b1) {
        uint64_t c = 0;
        for (int b = b0; b < b1; b++) c += hist_of(q, b);
        return c;
    };
    const int me = d->rank;
    const int* my_cut = cut.data() + (size_t) me * (rounds + 1);
    // landing offsets: group_off[j] = where group j of this rank's shard starts (also in the sorted result)
    std::vector<uint64_t> group_off(rounds + 1, 0);
    for (int j = 0; j < rounds; j++)
    {
        uint64_t c = 0;
        for (int src = 0; src < R; src++) c += count_in(src, my_cut[j], my_cut[j + 1]);
        group_off[j + 1] = group_off[j] + c;
    }
    std::vector<uint64_t> roff(R + 1, 0); // ordinary local sort: source-major, as with one round
    for (int r = 0; r < R; r++) roff[r + 1] = roff[r] + d->recv_counts[r];
    hipStream_t xs = rounds > 1 ? d->aux : st;
    if (rounds > 1)
    {
        HIP_TRY(hipEventRecord(d->ev_part, st)); // the partitioned slice is complete
        HIP_TRY(hipStreamWaitEvent(d->aux, d->ev_part, 0));
    }
    for (int j = 0; j < rounds; j++)
    {
        bool any_peer = false;
        for (int peer = 0; peer < R && !any_peer; peer++)
        {
            if (peer == me) continue;
            const int* pc = cut.data() + (size_t) peer * (rounds + 1);
            any_peer = count_in(me, pc[j], pc[j + 1]) != 0 || count_in(peer, my_cut[j], my_cut[j + 1]) != 0;
        }
        // where source `src`'s part of group j lands
        auto land_at = [&](int src) {
            uint64_t at;
            if (segmented)
            {
                at = group_off[j];
                for (int q = 0; q < src; q++) at += count_in(q, my_cut[j], my_cut[j + 1]);
            }
            else
            {
                at = roff[src];
                for (int jj = 0; jj < j; jj++) at += count_in(src, my_cut[jj], my_cut[jj + 1]);
            }
            return at;
        };
        if (any_peer)
        {
            NCCL_TRY(rccl().GroupStart());
            ncclResult_t res = ncclSuccess;
            for (int peer = 0; peer < R && res == ncclSuccess; peer++)
            {
                if (peer == me) continue;
                const int* pc = cut.data() + (size_t) peer * (rounds + 1);
                const uint64_t n_send = count_in(me, pc[j], pc[j + 1]), send_at = count_in(me, 0, pc[j]);
                const uint64_t n_get = count_in(peer, my_cut[j], my_cut[j + 1]), get_at = land_at(peer);
                if (n_send)
                {
                    res = rccl().Send(part_k + send_at, (size_t) n_send, ncclUint32, peer, d->comm, xs);
                    if (res == ncclSuccess) res = rccl().Send(part_v + send_at, (size_t) n_send, ncclUint32, peer, d->comm, xs);
                }
                if (n_get && res == ncclSuccess)
                {
                    res = rccl().Recv(land_k + get_at, (size_t) n_get, ncclUint32, peer, d->comm, xs);
                    if (res == ncclSuccess) res = rccl().Recv(land_v + get_at, (size_t) n_get, ncclUint32, peer, d->comm, xs);
                }
            }
            ncclResult_t end = rccl().GroupEnd();
            NCCL_TRY(res);
            NCCL_TRY(end);
        }
        // (fault injection, glu_dist_s::test_fault == 2: what follows the exchange runs on the side stream, not behind it)
        hipStream_t self_stream = (d->test_fault == 2 && rounds == 1) ? d->aux : xs;
        if (const uint64_t n_self = count_in(me, my_cut[j], my_cut[j + 1]))
        {
            const uint64_t from = count_in(me, 0, my_cut[j]), to = land_at(me);
            HIP_TRY(hipMemcpyAsync(land_k + to, part_k + from, (size_t) n_self * sizeof(uint32_t), hipMemcpyDeviceToDevice, self_stream));
            HIP_TRY(hipMemcpyAsync(land_v + to, part_v + from, (size_t) n_self * sizeof(uint32_t), hipMemcpyDeviceToDevice, self_stream));
        }
        if (rounds > 1) HIP_TRY(hipEventRecord(d->ev_round[j], xs));
    }
    glu_dist_s::Marks* marks = dist_marks(d, false);
    if (marks) HIP_TRY(hipEventRecord(marks->e[2], xs)); // (with rounds: the end of the last round, on the side stream)

    // 5. local stable sort of the received pairs
    hipStream_t sort_stream = (d->test_fault == 2 && rounds == 1) ? d->aux : st;
    if (segmented && rounds == 1)
        GLU_TRY(seg_run_plan(d->sorter, plan, land_k, land_v, recv_keys, recv_vals, sort_stream));
    else if (segmented)
    {
        // group by group, each behind its own round: the pieces of group j are (source, bucket) in source order per bucket,
        // at their places in the (16-byte aligned) landing arrays; its segments occupy [group_off[j], group_off[j + 1]) of the
        // result.  The sort kernels leave a few CUs to the RCCL kernels of the rounds still travelling (as the partition does
        // for the histogram exchange).
        const uint32_t saved_reserved = d->sorter->reserved_cus;
        const int leave = std::max(d->reserved_cus, 8);
        if (g_dev.num_cus > 4 * leave) d->sorter->reserved_cus = (uint32_t) leave;
        glu_status sorted = GLU_OK;
        for (int j = 0; j < rounds && sorted == GLU_OK; j++)
        {
            if (hipError_t e = hipStreamWaitEvent(st, d->ev_round[j], 0); e != hipSuccess)
            {
                sorted = fail(GLU_ERROR_DEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
                break;
            }
            const uint64_t count_j = group_off[j + 1] - group_off[j];
            if (count_j == 0) continue;
            std::vector<SegPiece> pieces;
            uint64_t at = group_off[j];
            for (int src = 0; src < R; src++)
                for (int bk = my_cut[j]; bk < my_cut[j + 1]; bk++)
                {
                    const uint64_t len = hist_of(src, bk);
                    if (len) pieces.push_back(SegPiece{at, len, (uint32_t) (bk - my_cut[j])});
                    at += len;
                }
            SegPlan gplan;
            seg_make_plan(d->sorter, std::move(pieces), (uint32_t) (my_cut[j + 1] - my_cut[j]), (size_t) count_j, 32 - kDistTopBits, true, gplan,
                          group_off[j]);
            sorted = seg_run_plan(d->sorter, gplan, land_k, land_v, recv_keys, recv_vals, st);
        }
        d->sorter->reserved_cus = saved_reserved;
        GLU_TRY(sorted);
    }
    else
    {
        if (rounds > 1) HIP_TRY(hipStreamWaitEvent(st, d->ev_round[rounds - 1], 0));
        if (n_recv > 1) GLU_TRY(sort_run<uint32_t>(d->sorter, recv_keys, recv_vals, n_recv, 0, sort_stream));
    }
    if (marks) HIP_TRY(hipEventRecord(marks->e[3], st));
    return GLU_OK;
}
} // namespace

extern "C" {

// Steps 4-5 into the caller's receive arrays (capacity in pairs, at least the count glu_dist_sort_begin returned).
glu_status glu_dist_sort_finish(glu_dist d, uint32_t* recv_keys, uint32_t* recv_vals, size_t capacity, void* stream)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    return dist_sort_finish(d, recv_keys, recv_vals, capacity, pick_stream(stream), GLU_OK);
}

// Both halves with the object's own receive arrays (grown when the shard does not fit: an allocation, so callers that
// must not allocate size them with glu_dist_prepare).  *out_keys / *out_vals stay valid until the next sort on `d`.
glu_status glu_dist_sort_ptr(glu_dist d, const uint32_t* keys, const uint32_t* vals, size_t local_count, void* stream,
                             uint32_t** out_keys, uint32_t** out_vals, size_t* out_count)
{
    size_t n_recv = 0;
    GLU_TRY(glu_dist_sort_begin(d, keys, vals, local_count, stream, &n_recv));
    // from here to the exchange a failure of this rank alone must not return: the ranks agree on it in dist_sort_finish
    glu_status grown = GLU_OK;
    const size_t have = d->recv_k.size / sizeof(uint32_t);
    // (the arrays a sort before the last one returned are released now: nobody may still hold them)
    d->retired_k.release();
    d->retired_v.release();
    if (have < std::max<size_t>(n_recv, 1))
    {
        // the exchange has not been posted yet: the partition may still be running on `stream`, the old arrays are idle.  They
        // are what the LAST sort returned (*out_keys / *out_vals, tensors of glu_hip.dist alias them): kept until the next call
        // instead of being freed under the caller's feet
        d->retired_k = d->recv_k;
        d->retired_v = d->recv_v;
        d->recv_k = Scratch();
        d->recv_v = Scratch();
        const size_t want = n_recv + n_recv / 8 + 4096;
        grown = d->recv_k.reserve(want * sizeof(uint32_t));
        if (grown == GLU_OK) grown = d->recv_v.reserve(want * sizeof(uint32_t));
    }
    const size_t capacity = grown == GLU_OK ? std::min(d->recv_k.size, d->recv_v.size) / sizeof(uint32_t) : 0;
    GLU_TRY(dist_sort_finish(d, (uint32_t*) d->recv_k.ptr, (uint32_t*) d->recv_v.ptr, capacity, pick_stream(stream), grown));
    if (out_keys) *out_keys = (uint32_t*) d->recv_k.ptr;
    if (out_vals) *out_vals = (uint32_t*) d->recv_v.ptr;
    if (out_count) *out_count = n_recv;
    return GLU_OK;
}

glu_status glu_dist_set_rounds(glu_dist d, int rounds)
{
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (rounds < 1 || rounds > glu_dist_s::kMaxRounds) return fail(GLU_ERROR_INVALID_ARGUMENT, "rounds must be 1 .. %d (got %d)", glu_dist_s::kMaxRounds, rounds);
    d->rounds = rounds; // (every rank must set the same value: the rounds are part of the message sequence)
    return GLU_OK;
}

glu_status glu_dist_last_rounds(glu_dist d, uint32_t* rounds)
{
    if (!d || !rounds) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *rounds = d->last_rounds;
    return GLU_OK;
}

glu_status glu_dist_last_local_sort(glu_dist d, uint32_t* segmented)
{
    if (!d || !segmented) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL argument");
    *segmented = d->last_local_sort;
    return GLU_OK;
}

} // extern "C"
