// mock_rccl.cpp -- TEST DOUBLE for the nine librccl entry points glu_dist_* binds (gl-radix-sort_amd/csrc/glu_dist_impl.hpp).
//
// RCCL refuses two ranks on one GPU, and the GPU box of this project has one GPU: the multi-rank code of the sharded sort
// (plan with world > 1, send / receive offsets, receive order, empty shards, the lower-byte fallback agreed between
// ranks) would never run before the 8-GPU bench does.  This library lets several PROCESSES sharing one GPU act as ranks:
// GLU_HIP_RCCL_LIB=<this .so> makes libglu_hip.so bind it instead of librccl; the ranks exchange through files in
// $GLU_MOCK_RCCL_DIR (device -> host -> file -> host -> device).  It is not a transport anybody should ship.  Two modes:
//   * default (synchronous): every call synchronises the stream and blocks on the host.  That proves offsets, counts and the
//     plan, but it HIDES stream-order bugs: whatever the caller forgot to order before the collective has finished anyway.
//   * GLU_MOCK_RCCL_ASYNC=1 (asynchronous, like the real library): a call only ENQUEUES.  It records an event on the
//     caller's stream, launches a one-lane kernel there that waits for a flag in pinned host memory, hands the work to the
//     communicator's worker thread and returns at once; ncclGroupEnd does not wait either.  What is sent is copied into pinned
//     buffers by the caller's stream itself, in stream order, in front of the event; the worker waits for the event (= the
//     stream has reached the collective), moves the bytes between pinned buffers and files, then raises the flag, and the
//     caller's stream runs on into the copies of what was received (enqueued with the call, behind the wait kernel).  A send buffer that is not yet written when the stream gets there, a
//     receive buffer still in use, a histogram gathered before its kernel ran: the data is wrong and the test fails, as it
//     would (sometimes) with RCCL.  Errors of the worker surface at the next call on the communicator.
// What both keep of the real semantics is what the caller relies on:
//   * ncclAllGather: rank r's `count` elements land at recv + r * count on every rank;
//   * ncclSend / ncclRecv inside ncclGroupStart .. ncclGroupEnd: the i-th send of rank a to rank b pairs with the i-th
//     receive of rank b from rank a; a size mismatch is an error (real RCCL would hang or corrupt: here the test fails);
//   * work enqueued on the stream after the call sees the received data.
// Used only by tests/test_gpu_dist.py::test_native_multi_rank_*.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace
{
struct MockComm;

struct Op
{
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    MockComm* comm;
    hipStream_t stream;
    std::string path;           // asynchronous mode: named when the call is made (call order pairs sends with receives)
    const void* gather_send = nullptr; // asynchronous all-gather: ptr = receive array, bytes per rank
    bool gather = false;
    unsigned char* out_stage = nullptr; // asynchronous mode: pinned copies of what leaves / arrives (see enqueue_ops)
    unsigned char* in_stage = nullptr;
};

struct Stage // a pinned host buffer of the communicator's pool
{
    unsigned char* host = nullptr;
    size_t capacity = 0;
    bool busy = false;
};

struct Job
{
    std::vector<Op> ops;
    std::vector<hipEvent_t> ready; // one per stream: everything the operations send has reached its pinned copy
    std::vector<hipEvent_t> done;  // one per stream: everything received has been copied to its destination
    std::vector<Stage*> stages;
    uint32_t seq = 0;
};

struct MockComm
{
    int nranks = 1, rank = 0;
    std::string tag;
    uint64_t gather_seq = 0;
    std::vector<uint64_t> send_seq, recv_seq; // per peer
    // asynchronous mode
    bool async = false;
    int device = 0;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool stop = false;
    uint32_t* flag = nullptr; // pinned, coherent: number of jobs the worker has completed
    uint32_t enqueued = 0;
    std::atomic<int> failed{(int) ncclSuccess};
    std::vector<Stage*> pool;  // pinned staging buffers, grow-only, freed with the communicator (calling thread only)
    std::deque<Job> retired;   // jobs the worker has finished; their buffers are free again once their `done` events have fired
};

// the caller's stream stops here until the worker has finished job `target` (or gave up: the worker always raises the flag;
// the bound only keeps a dead worker from hanging the device)
__global__ void mock_wait_kernel(const uint32_t* flag, uint32_t target)
{
    const uint64_t t0 = wall_clock64();
    while ((int32_t) (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0)
    {
        if (wall_clock64() - t0 > 150ull * 100000000ull) break; // 150 s of the 100 MHz wall clock
        __builtin_amdgcn_s_sleep(127);
    }
}

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

const char* dir()
{
    const char* d = getenv("GLU_MOCK_RCCL_DIR");
    return d && *d ? d : "/tmp";
}

bool verbose() // GLU_MOCK_RCCL_VERBOSE: every operation is logged to stderr
{
    static const bool v = getenv("GLU_MOCK_RCCL_VERBOSE") != nullptr;
    return v;
}

size_t type_size(ncclDataType_t t)
{
    switch (t)
    {
    case ncclInt8:
    case ncclUint8: return 1;
    case ncclFloat16:
    case ncclBfloat16: return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32: return 4;
    case ncclInt64:
    case ncclUint64:
    case ncclFloat64: return 8;
    default: return 0;
    }
}

bool write_file(const std::string& path, const void* data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0; // appears atomically
}

// waits for the file (another rank writes it), checks its size, reads it
ncclResult_t read_file(const std::string& path, void* data, size_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    const char* limit_env = getenv("GLU_MOCK_RCCL_TIMEOUT_S");
    const int limit_s = limit_env && atoi(limit_env) > 0 ? atoi(limit_env) : 120;
    struct stat st;
    while (stat(path.c_str(), &st) != 0)
    {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit_s))
        {
            fprintf(stderr, "[mock_rccl] timed out waiting for %s\n", path.c_str());
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if ((size_t) st.st_size != bytes)
    {
        fprintf(stderr, "[mock_rccl] %s holds %zu bytes, the receiver expects %zu\n", path.c_str(), (size_t) st.st_size, bytes);
        return ncclInvalidArgument;
    }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return ncclSystemError;
    const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok ? ncclSuccess : ncclSystemError;
}

std::string p2p_path(MockComm* c, bool send, int peer)
{
    return send ? std::string(dir()) + "/" + c->tag + ".p2p." + std::to_string(c->rank) + "." + std::to_string(peer) + "." +
                      std::to_string(c->send_seq[peer]++)
                : std::string(dir()) + "/" + c->tag + ".p2p." + std::to_string(peer) + "." + std::to_string(c->rank) + "." +
                      std::to_string(c->recv_seq[peer]++);
}

void log_row(const MockComm* c, const std::string& base, int r, const unsigned char* row, size_t bytes)
{
    if (!verbose() || bytes < 1028) return;
    uint64_t sum = 0;
    for (size_t i = 0; i < 256; i++) sum += reinterpret_cast<const uint32_t*>(row)[i];
    fprintf(stderr, "[mock_rccl %d] %s row of rank %d: first 256 words add up to %llu, word 256 = %u\n", c->rank,
            base.c_str() + base.rfind('/') + 1, r, (unsigned long long) sum, reinterpret_cast<const uint32_t*>(row)[256]);
}

// ---- synchronous mode: on the calling thread, after the caller's streams were synchronised ---------------------------
ncclResult_t gather_now(MockComm* c, const void* send, void* recv, size_t bytes, const std::string& base)
{
    std::vector<unsigned char> host(bytes);
    if (bytes && hipMemcpy(host.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!write_file(base + std::to_string(c->rank), host.data(), bytes)) return ncclSystemError;
    for (int r = 0; r < c->nranks; r++)
    {
        if (ncclResult_t res = read_file(base + std::to_string(r), host.data(), bytes); res != ncclSuccess) return res;
        log_row(c, base, r, host.data(), bytes);
        if (bytes && hipMemcpy((unsigned char*) recv + (size_t) r * bytes, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t move_ops_now(std::vector<Op>& ops)
{
    std::vector<unsigned char> host;
    for (const Op& op : ops) // all sends first: nobody waits for a peer before its own data is out
    {
        if (!op.send) continue;
        host.resize(op.bytes);
        if (op.bytes && hipMemcpy(host.data(), op.ptr, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!write_file(op.path, host.data(), op.bytes)) return ncclSystemError;
    }
    for (const Op& op : ops)
    {
        if (op.send) continue;
        host.resize(op.bytes);
        if (ncclResult_t r = read_file(op.path, host.data(), op.bytes); r != ncclSuccess) return r;
        unlink(op.path.c_str());
        if (op.bytes && hipMemcpy(op.ptr, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

// ---- asynchronous mode --------------------------------------------------------------------------------------------------
// Everything the GPU does for a collective is enqueued BY THE CALLER'S THREAD ON THE CALLER'S STREAM when the call is made:
//     copies of what is sent into pinned buffers -> event `ready` -> mock_wait_kernel -> copies of what is received out of
//     pinned buffers -> event `done`
// and the worker thread only waits for `ready` on the host, moves bytes between pinned buffers and files, and raises the flag.
// (A first version let the worker copy on a private stream.  HIP multiplexes its streams onto a few hardware queues, in
// order: the worker's copy could land in the queue behind the very wait kernel that was waiting for it.)  What shares a
// hardware queue with a parked stream is delayed, as behind an RCCL kernel that waits for its peers; nothing the worker
// needs is ever behind a wait kernel of a LATER operation, because a rank issues its operations in one order.
void worker_main(MockComm* c)
{
    (void) hipSetDevice(c->device);
    for (;;)
    {
        Job job;
        {
            std::unique_lock<std::mutex> lock(c->m);
            c->cv.wait(lock, [&] { return c->stop || !c->jobs.empty(); });
            if (c->jobs.empty()) return;
            job = std::move(c->jobs.front());
            c->jobs.pop_front();
        }
        ncclResult_t res = ncclSuccess;
        for (hipEvent_t e : job.ready) // the caller's stream has reached the collective: what it sends is what it is NOW
            if (hipEventSynchronize(e) != hipSuccess) res = ncclUnhandledCudaError;
        for (const Op& op : job.ops) // all sends (and own gather rows) first: nobody waits for a peer before its data is out
        {
            if (res != ncclSuccess) break;
            if (op.gather)
            {
                if (!write_file(op.path + std::to_string(c->rank), op.out_stage, op.bytes)) res = ncclSystemError;
            }
            else if (op.send && !write_file(op.path, op.out_stage, op.bytes))
                res = ncclSystemError;
        }
        for (const Op& op : job.ops)
        {
            if (res != ncclSuccess) break;
            if (op.gather)
            {
                for (int r = 0; r < c->nranks && res == ncclSuccess; r++)
                {
                    res = read_file(op.path + std::to_string(r), op.in_stage + (size_t) r * op.bytes, op.bytes);
                    if (res == ncclSuccess) log_row(c, op.path, r, op.in_stage + (size_t) r * op.bytes, op.bytes);
                }
            }
            else if (!op.send)
            {
                res = read_file(op.path, op.in_stage, op.bytes);
                if (res == ncclSuccess) unlink(op.path.c_str());
            }
        }
        if (res != ncclSuccess)
        {
            fprintf(stderr, "[mock_rccl] rank %d: asynchronous operation %u failed (%d)\n", c->rank, job.seq, (int) res);
            c->failed.store((int) res);
        }
        if (verbose()) fprintf(stderr, "[mock_rccl %d] operation %u of %s: bytes moved, stream released\n", c->rank, job.seq, c->tag.c_str());
        const uint32_t seq = job.seq;
        {
            std::lock_guard<std::mutex> lock(c->m);
            c->retired.push_back(std::move(job));
        }
        __atomic_store_n(c->flag, seq, __ATOMIC_RELEASE); // the caller's stream runs on: the copies out of the pinned buffers
    }
}

// calling thread: buffers of jobs that are finished on both sides (worker and stream) return to the pool
void collect_retired(MockComm* c)
{
    std::lock_guard<std::mutex> lock(c->m);
    while (!c->retired.empty())
    {
        Job& j = c->retired.front();
        bool fired = true;
        for (hipEvent_t e : j.done) fired = fired && hipEventQuery(e) == hipSuccess;
        if (!fired) break;
        for (hipEvent_t e : j.ready) (void) hipEventDestroy(e);
        for (hipEvent_t e : j.done) (void) hipEventDestroy(e);
        for (Stage* st : j.stages) st->busy = false;
        c->retired.pop_front();
    }
}

unsigned char* acquire_stage(MockComm* c, Job& job, size_t bytes)
{
    if (bytes == 0) return nullptr;
    Stage* pick = nullptr;
    for (Stage* st : c->pool)
        if (!st->busy && st->capacity >= bytes && (!pick || st->capacity < pick->capacity)) pick = st;
    if (!pick)
    {
        pick = new Stage();
        pick->capacity = std::max<size_t>(bytes, 4096);
        if (hipHostMalloc((void**) &pick->host, pick->capacity, hipHostMallocDefault) != hipSuccess)
        {
            delete pick;
            return nullptr;
        }
        c->pool.push_back(pick);
    }
    pick->busy = true;
    job.stages.push_back(pick);
    return pick->host;
}

ncclResult_t enqueue_ops(MockComm* c, std::vector<Op>& ops)
{
    if (int f = c->failed.load(); f != (int) ncclSuccess) return (ncclResult_t) f;
    collect_retired(c);
    Job job;
    job.seq = ++c->enqueued;
    std::vector<hipStream_t> streams;
    for (const Op& op : ops)
    {
        bool seen = false;
        for (hipStream_t s : streams) seen = seen || s == op.stream;
        if (!seen) streams.push_back(op.stream);
    }
    for (Op& op : ops) // 1. what leaves: device -> pinned copy, in stream order
    {
        const void* src = op.gather ? op.gather_send : (op.send ? op.ptr : nullptr);
        if (!src || !op.bytes) continue;
        op.out_stage = acquire_stage(c, job, op.bytes);
        if (!op.out_stage || hipMemcpyAsync(op.out_stage, src, op.bytes, hipMemcpyDeviceToHost, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (hipStream_t s : streams) // 2. `ready`, then the stream parks until the worker has moved the bytes
    {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, s) != hipSuccess) return ncclUnhandledCudaError;
        job.ready.push_back(e);
        hipLaunchKernelGGL(mock_wait_kernel, dim3(1), dim3(1), 0, s, c->flag, job.seq);
        if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    }
    for (Op& op : ops) // 3. what arrives: pinned copy -> device, behind the wait kernel
    {
        if (op.send && !op.gather) continue;
        const size_t bytes = op.gather ? op.bytes * (size_t) c->nranks : op.bytes;
        if (!bytes) continue;
        op.in_stage = acquire_stage(c, job, bytes);
        if (!op.in_stage || hipMemcpyAsync(op.ptr, op.in_stage, bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (hipStream_t s : streams)
    {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, s) != hipSuccess) return ncclUnhandledCudaError;
        job.done.push_back(e);
    }
    if (verbose()) fprintf(stderr, "[mock_rccl %d] operation %u of %s enqueued\n", c->rank, job.seq, c->tag.c_str());
    job.ops = std::move(ops);
    {
        std::lock_guard<std::mutex> lock(c->m);
        c->jobs.push_back(std::move(job));
    }
    c->cv.notify_one();
    return ncclSuccess;
}

ncclResult_t run_ops(std::vector<Op>& ops)
{
    if (ops.empty()) return ncclSuccess;
    for (Op& op : ops) op.path = p2p_path(op.comm, op.send, op.peer);
    if (verbose())
        for (const Op& op : ops)
            fprintf(stderr, "[mock_rccl %d] %s %zu bytes %s %d as %s\n", op.comm->rank, op.send ? "send" : "recv", op.bytes, op.send ? "to" : "from",
                    op.peer, op.path.c_str() + op.path.rfind('/') + 1);
    if (ops.front().comm->async) return enqueue_ops(ops.front().comm, ops);
    for (const Op& op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    return move_ops_now(ops);
}

ncclResult_t p2p(bool send, void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (!c || peer < 0 || peer >= c->nranks || type_size(type) == 0) return ncclInvalidArgument;
    if (count && !ptr) return ncclInvalidArgument;
    Op op{send, ptr, count * type_size(type), peer, c, stream, std::string()};
    if (g_depth > 0)
    {
        g_ops.push_back(op);
        return ncclSuccess;
    }
    std::vector<Op> one{op};
    return run_ops(one);
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    static int counter = 0;
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof(id->internal), "mock-%d-%lld-%d", (int) getpid(), (long long) now, counter++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    MockComm* c = new MockComm();
    c->nranks = nranks;
    c->rank = rank;
    char hex[2 * 24 + 1]; // the first 24 bytes of the id, whatever they are, name the communicator's files
    for (int i = 0; i < 24; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned) (unsigned char) id.internal[i]);
    c->tag = hex;
    c->send_seq.assign(nranks, 0);
    c->recv_seq.assign(nranks, 0);
    if (const char* e = getenv("GLU_MOCK_RCCL_ASYNC"); e && atoi(e) != 0)
    {
        c->async = true;
        if (hipGetDevice(&c->device) != hipSuccess ||
            hipHostMalloc((void**) &c->flag, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
        {
            delete c;
            return ncclUnhandledCudaError;
        }
        *c->flag = 0;
        c->worker = std::thread(worker_main, c);
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (c && c->async)
    {
        {
            std::lock_guard<std::mutex> lock(c->m);
            c->stop = true; // (the worker finishes what is queued first)
        }
        c->cv.notify_one();
        c->worker.join();
        (void) hipDeviceSynchronize(); // (every stream has passed its wait kernels: the worker finished all jobs)
        for (Job& j : c->retired)
        {
            for (hipEvent_t e : j.ready) (void) hipEventDestroy(e);
            for (hipEvent_t e : j.done) (void) hipEventDestroy(e);
        }
        for (Stage* st : c->pool)
        {
            (void) hipHostFree(st->host);
            delete st;
        }
        (void) hipHostFree(c->flag);
    }
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r)
    {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "mock: HIP call failed";
    case ncclSystemError: return "mock: file exchange failed or timed out";
    case ncclInvalidArgument: return "mock: invalid argument (or send / receive sizes disagree)";
    default: return "mock: error";
    }
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    const size_t bytes = count * type_size(type);
    if (!c || type_size(type) == 0 || (bytes && (!send || !recv))) return ncclInvalidArgument;
    const std::string base = std::string(dir()) + "/" + c->tag + ".gather." + std::to_string(c->gather_seq++) + ".";
    if (verbose()) fprintf(stderr, "[mock_rccl %d] all-gather %zu bytes as %s\n", c->rank, bytes, base.c_str() + base.rfind('/') + 1);
    if (c->async)
    {
        Op op{false, recv, bytes, c->rank, c, stream, base};
        op.gather = true;
        op.gather_send = send;
        std::vector<Op> one{op};
        return enqueue_ops(c, one);
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return gather_now(c, send, recv, bytes, base);
}

ncclResult_t ncclSend(const void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(true, const_cast<void*>(ptr), count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(false, ptr, count, type, peer, comm, stream);
}

ncclResult_t ncclGroupStart()
{
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

} // extern "C"
