// glu_dist_impl.hpp -- the sharded (multi-GPU) radix sort behind the glu_dist_* entry points of include/glu_hip.h.
// Included by glu_hip.hip (same translation unit: it uses the library's pass launchers and scratch objects).
//
// The reference is single-device (replicas only); BASELINE.json configs[3] asks for the
// sharded form: rank r holds slice r of the array, one exchange step on the top 8 key bits, then local sorts.
//   1. stable partition of the local slice by the bucket key >> 24 with the sort's own count / scan / scatter kernels
//      (the 256-bin bucket histogram falls out of the row scan, before the scatter runs);
//   2. ncclAllGather of the R x 256 histograms and the copy to the host run on a side stream while the partition's
//      scatter kernel is still running (it leaves two CUs free for the RCCL kernel): the host has the plan by the time
//      the partition is done and posts the exchange right behind it, the sort's stream never waits for the host;
//   3. every rank derives the same contiguous bucket -> rank map (a bucket is never split) and its send / receive counts;
//   4. ONE grouped exchange (ncclGroupStart .. ncclGroupEnd) carries keys and values to and from every peer, receive
//      segments in source-rank order; the part that stays on the rank is a device copy;
//   5. local stable sort of what was received.
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
        const char* names[] = {getenv("GLU_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
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
    Scratch hist;                       // [256] local bucket histogram, [world * 256] gathered
    uint32_t* all_hist_host = nullptr;  // pinned
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
    if (const char* e = getenv("GLU_HIP_DIST_TEST_REPARTITION")) d->repartition_at_world_1 = atoi(e) != 0;
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
    if (e == hipSuccess) e = hipHostMalloc((void**) &d->all_hist_host, (size_t) world_size * kDistBuckets * sizeof(uint32_t));
    if (e != hipSuccess) return cleanup(fail(GLU_ERROR_DEVICE, "glu_dist_create: %s", hipGetErrorString(e)));
    if (glu_status st = d->hist.reserve((size_t) (world_size + 1) * kDistBuckets * sizeof(uint32_t)); st != GLU_OK) return cleanup(st);
    *out = d;
    return GLU_OK;
}

glu_status glu_dist_destroy(glu_dist d)
{
    if (!d) return GLU_OK;
    (void) hipDeviceSynchronize();
    if (d->comm) (void) rccl().CommDestroy(d->comm);
    if (d->sorter) (void) glu_radix_sort_destroy(d->sorter);
    if (d->aux) (void) hipStreamDestroy(d->aux);
    if (d->ev_hist) (void) hipEventDestroy(d->ev_hist);
    if (d->ev_plan) (void) hipEventDestroy(d->ev_plan);
    for (glu_dist_s::Marks& m : d->marks)
        for (hipEvent_t e : m.e)
            if (e) (void) hipEventDestroy(e);
    if (d->all_hist_host) (void) hipHostFree(d->all_hist_host);
    d->part_k.release();
    d->part_v.release();
    d->recv_k.release();
    d->recv_v.release();
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
    GLU_TRY(d->part_k.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    GLU_TRY(d->part_v.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    if (recv_capacity)
    {
        GLU_TRY(d->recv_k.reserve(recv_capacity * sizeof(uint32_t)));
        GLU_TRY(d->recv_v.reserve(recv_capacity * sizeof(uint32_t)));
    }
    // partition pass: table only; local sort: scratch for the receive side
    GLU_TRY(sort_prepare(d->sorter, std::max(local_count, recv_capacity), sizeof(uint32_t), true));
    return GLU_OK;
}

glu_status glu_dist_set_reserved_cus(glu_dist d, int cus)
{
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (cus < 0 || cus * 4 >= g_dev.num_cus) return fail(GLU_ERROR_INVALID_ARGUMENT, "cannot reserve %d of %d CUs", cus, g_dev.num_cus);
    d->reserved_cus = cus;
    d->sorter->max_blocks = cus ? (uint32_t) (g_dev.num_cus - cus) : 0u;
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

// Steps 1-3: partition the local slice, exchange the histograms, plan.  The partition is enqueued on `stream`; the call
// returns when the host has the plan (it waited for the histogram exchange on the side stream, not for the partition's
// scatter kernel).  *recv_count = number of pairs this rank will receive: the caller sizes its receive arrays with it.
glu_status glu_dist_sort_begin(glu_dist d, const uint32_t* keys, const uint32_t* vals, size_t local_count, void* stream,
                               size_t* recv_count)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (local_count > 0 && (!keys || !vals)) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL array");
    if (local_count > 0xFFFF0000ull) return fail(GLU_ERROR_INVALID_ARGUMENT, "count %zu does not fit 32-bit indexing", local_count);
    hipStream_t st = pick_stream(stream);
    glu_dist_s::Marks* marks = dist_marks(d, true);
    GLU_TRY(d->part_k.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    GLU_TRY(d->part_v.reserve(std::max<size_t>(local_count, 1) * sizeof(uint32_t)));
    uint32_t* hist = (uint32_t*) d->hist.ptr;
    uint32_t* all_hist = hist + kDistBuckets;
    if (marks) HIP_TRY(hipEventRecord(marks->e[0], st));

    // 1 + 2, on the top key byte first.  If every key of every rank shares that byte (24-bit keys, small integers ...) all
    // the data would land on one rank: the partition is repeated on the next lower byte, down to the lowest.  The bytes
    // above the chosen one are constant over the whole input, so bucket order is still key order and a rank's shard is a
    // contiguous key range.  (The first attempt was then an identity copy of the slice: 0.6 ms at 2^27, paid only by such
    // inputs.)  Every rank sees the same gathered histograms and takes the same decision.
    for (uint32_t shift = 32 - kDistTopBits;; shift -= kDistTopBits)
    {
        // 1. stable partition by the bucket (key >> shift) & 255; the histogram is ready (and ev_hist recorded) after the
        //    row scan, before the scatter
        if (local_count == 0)
        {
            HIP_TRY(hipMemsetAsync(hist, 0, kDistBuckets * sizeof(uint32_t), st));
            HIP_TRY(hipEventRecord(d->ev_hist, st));
        }
        else
        {
            uint64_t cap = (uint64_t) g_dev.num_cus * kMaxBlocksPerCu;
            GLU_TRY(d->sorter->table.reserve((kMaxRadix * cap + kMaxRadix) * sizeof(uint32_t)));
            const uint32_t saved_blocks = d->sorter->max_blocks;
            // The histogram exchange of step 2 needs a CU beside the scatter, and workgroups are dealt round-robin to the 8
            // XCDs: with 254 workgroups six XCDs are full and a one-workgroup kernel bound for one of them waits for the
            // scatter to end (measured: 0.40 ms for the exchange with 2 CUs left, 0.016 ms with 8 = one per XCD).
            const int leave = std::max(d->reserved_cus, 8);
            if (leave > 0 && g_dev.num_cus > 4 * leave && (saved_blocks == 0 || saved_blocks > (uint32_t) (g_dev.num_cus - leave)))
                d->sorter->max_blocks = (uint32_t) (g_dev.num_cus - leave);
            d->sorter->after_histogram_event = d->ev_hist;
            glu_status ps = dispatch_pass<uint32_t>(d->sorter, keys, vals, (uint32_t*) d->part_k.ptr, (uint32_t*) d->part_v.ptr,
                                                    local_count, shift, kDistTopBits, hist, st);
            d->sorter->after_histogram_event = nullptr;
            d->sorter->max_blocks = saved_blocks;
            GLU_TRY(ps);
        }
        if (marks && shift == 32 - kDistTopBits) HIP_TRY(hipEventRecord(marks->e[1], st));

        // 2. every rank learns every rank's histogram (R x 256 words): side stream, beside the scatter kernel
        HIP_TRY(hipStreamWaitEvent(d->aux, d->ev_hist, 0));
        if (marks && shift == 32 - kDistTopBits) HIP_TRY(hipEventRecord(marks->e[4], d->aux));
        NCCL_TRY(rccl().AllGather(hist, all_hist, kDistBuckets, ncclUint32, d->comm, d->aux));
        HIP_TRY(hipMemcpyAsync(d->all_hist_host, all_hist, (size_t) d->world * kDistBuckets * sizeof(uint32_t), hipMemcpyDeviceToHost, d->aux));
        if (marks) HIP_TRY(hipEventRecord(marks->e[5], d->aux));
        HIP_TRY(hipEventRecord(d->ev_plan, d->aux));
        HIP_TRY(hipEventSynchronize(d->ev_plan));
        d->partition_shift = shift;
        if (shift == 0 || (d->world == 1 && !d->repartition_at_world_1) || !dist_single_bucket(d->all_hist_host, d->world)) break;
        if (marks) HIP_TRY(hipEventRecord(marks->e[1], st)); // the repeated partition counts as partition time
    }

    // 3. identical plan on every rank
    dist_plan_buckets(d->all_hist_host, d->world, d->owner.data());
    dist_plan_counts(d->all_hist_host, d->world, d->rank, d->owner.data(), d->send_counts.data(), d->recv_counts.data());
    uint64_t total = 0, sent = 0;
    for (int r = 0; r < d->world; r++) total += d->recv_counts[r], sent += d->send_counts[r];
    if (sent != local_count) return fail(GLU_ERROR_INVALID_STATE, "bucket histogram sums to %llu, expected %zu", (unsigned long long) sent, local_count);
    if (total > 0xFFFF0000ull)
        return fail(GLU_ERROR_INVALID_ARGUMENT,
                    "this rank would receive %llu pairs (buckets are never split: a hot bucket bounds the balance), more than one device sorts",
                    (unsigned long long) total);
    d->local_count = local_count;
    d->recv_total = total;
    d->began = true;
    if (recv_count) *recv_count = (size_t) total;
    return GLU_OK;
}

// Steps 4-5 into the caller's receive arrays (capacity in pairs, at least the count glu_dist_sort_begin returned).
glu_status glu_dist_sort_finish(glu_dist d, uint32_t* recv_keys, uint32_t* recv_vals, size_t capacity, void* stream)
{
    GLU_TRY(enter());
    if (!d) return fail(GLU_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (!d->began) return fail(GLU_ERROR_INVALID_STATE, "glu_dist_sort_finish without glu_dist_sort_begin");
    d->began = false;
    const size_t n_recv = (size_t) d->recv_total;
    if (n_recv > capacity) return fail(GLU_ERROR_INVALID_ARGUMENT, "receive arrays hold %zu pairs, %zu arrive", capacity, n_recv);
    if (n_recv > 0 && (!recv_keys || !recv_vals)) return fail(GLU_ERROR_INVALID_ARGUMENT, "NULL receive array");
    hipStream_t st = pick_stream(stream);
    const uint32_t* part_k = (const uint32_t*) d->part_k.ptr;
    const uint32_t* part_v = (const uint32_t*) d->part_v.ptr;

    // 4. one grouped exchange: keys and values to and from every peer; receive segments in source-rank order
    std::vector<uint64_t> soff(d->world + 1, 0), roff(d->world + 1, 0);
    for (int r = 0; r < d->world; r++) soff[r + 1] = soff[r] + d->send_counts[r], roff[r + 1] = roff[r] + d->recv_counts[r];
    bool any_peer = false;
    for (int r = 0; r < d->world; r++) any_peer = any_peer || (r != d->rank && (d->send_counts[r] || d->recv_counts[r]));
    if (any_peer)
    {
        NCCL_TRY(rccl().GroupStart());
        ncclResult_t res = ncclSuccess;
        for (int peer = 0; peer < d->world && res == ncclSuccess; peer++)
        {
            if (peer == d->rank) continue;
            if (d->send_counts[peer])
            {
                res = rccl().Send(part_k + soff[peer], (size_t) d->send_counts[peer], ncclUint32, peer, d->comm, st);
                if (res == ncclSuccess) res = rccl().Send(part_v + soff[peer], (size_t) d->send_counts[peer], ncclUint32, peer, d->comm, st);
            }
            if (d->recv_counts[peer] && res == ncclSuccess)
            {
                res = rccl().Recv(recv_keys + roff[peer], (size_t) d->recv_counts[peer], ncclUint32, peer, d->comm, st);
                if (res == ncclSuccess) res = rccl().Recv(recv_vals + roff[peer], (size_t) d->recv_counts[peer], ncclUint32, peer, d->comm, st);
            }
        }
        ncclResult_t end = rccl().GroupEnd();
        NCCL_TRY(res);
        NCCL_TRY(end);
    }
    if (d->send_counts[d->rank])
    {
        const size_t bytes = (size_t) d->send_counts[d->rank] * sizeof(uint32_t);
        HIP_TRY(hipMemcpyAsync(recv_keys + roff[d->rank], part_k + soff[d->rank], bytes, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(recv_vals + roff[d->rank], part_v + soff[d->rank], bytes, hipMemcpyDeviceToDevice, st));
    }
    glu_dist_s::Marks* marks = dist_marks(d, false);
    if (marks) HIP_TRY(hipEventRecord(marks->e[2], st));

    // 5. local stable sort of the received pairs
    if (n_recv > 1) GLU_TRY(sort_run<uint32_t>(d->sorter, recv_keys, recv_vals, n_recv, 0, st));
    if (marks) HIP_TRY(hipEventRecord(marks->e[3], st));
    return GLU_OK;
}

// Both halves with the object's own receive arrays (grown when the shard does not fit: an allocation, so callers that
// must not allocate size them with glu_dist_prepare).  *out_keys / *out_vals stay valid until the next sort on `d`.
glu_status glu_dist_sort_ptr(glu_dist d, const uint32_t* keys, const uint32_t* vals, size_t local_count, void* stream,
                             uint32_t** out_keys, uint32_t** out_vals, size_t* out_count)
{
    size_t n_recv = 0;
    GLU_TRY(glu_dist_sort_begin(d, keys, vals, local_count, stream, &n_recv));
    const size_t have = d->recv_k.size / sizeof(uint32_t);
    if (have < n_recv)
    {
        // the exchange has not been posted yet: the partition may still be running on `stream`, the old arrays are idle
        const size_t want = n_recv + n_recv / 8 + 4096;
        GLU_TRY(d->recv_k.reserve(want * sizeof(uint32_t)));
        GLU_TRY(d->recv_v.reserve(want * sizeof(uint32_t)));
    }
    GLU_TRY(sort_prepare(d->sorter, n_recv, sizeof(uint32_t), true));
    GLU_TRY(glu_dist_sort_finish(d, (uint32_t*) d->recv_k.ptr, (uint32_t*) d->recv_v.ptr, d->recv_k.size / sizeof(uint32_t), stream));
    if (out_keys) *out_keys = (uint32_t*) d->recv_k.ptr;
    if (out_vals) *out_vals = (uint32_t*) d->recv_v.ptr;
    if (out_count) *out_count = n_recv;
    return GLU_OK;
}

} // extern "C"
