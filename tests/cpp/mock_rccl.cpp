// mock_rccl.cpp -- TEST DOUBLE for the nine librccl entry points glu_dist_* binds (gl-radix-sort_amd/csrc/glu_dist_impl.hpp).
//
// RCCL refuses two ranks on one GPU, and the GPU box of this project has one GPU: the multi-rank code of the sharded sort
// (plan with world > 1, send / receive offsets, receive order, empty shards, the lower-byte fallback agreed between
// ranks) would never run before the 8-GPU bench does.  This library lets several PROCESSES sharing one GPU act as ranks:
// GLU_HIP_RCCL_LIB=<this .so> makes libglu_hip.so bind it instead of librccl; the ranks exchange through files in
// $GLU_MOCK_RCCL_DIR (device -> host -> file -> host -> device).  It is not a transport anybody should ship: every call
// synchronises the stream and blocks on the host.  What it keeps of the real semantics is what the caller relies on:
//   * ncclAllGather: rank r's `count` elements land at recv + r * count on every rank;
//   * ncclSend / ncclRecv inside ncclGroupStart .. ncclGroupEnd: the i-th send of rank a to rank b pairs with the i-th
//     receive of rank b from rank a; a size mismatch is an error (real RCCL would hang or corrupt: here the test fails);
//   * work enqueued on the stream after the call sees the received data.
// Used only by tests/test_gpu_dist.py::test_native_multi_rank_*.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace
{
struct MockComm
{
    int nranks = 1, rank = 0;
    std::string tag;
    uint64_t gather_seq = 0;
    std::vector<uint64_t> send_seq, recv_seq; // per peer
};

struct Op
{
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    MockComm* comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

const char* dir()
{
    const char* d = getenv("GLU_MOCK_RCCL_DIR");
    return d && *d ? d : "/tmp";
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
    struct stat st;
    while (stat(path.c_str(), &st) != 0)
    {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
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

ncclResult_t run_ops(std::vector<Op>& ops)
{
    for (const Op& op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<unsigned char> host;
    for (const Op& op : ops) // all sends first: nobody waits for a peer before its own data is out
    {
        if (!op.send) continue;
        host.resize(op.bytes);
        if (op.bytes && hipMemcpy(host.data(), op.ptr, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        const std::string path = std::string(dir()) + "/" + op.comm->tag + ".p2p." + std::to_string(op.comm->rank) + "." +
                                 std::to_string(op.peer) + "." + std::to_string(op.comm->send_seq[op.peer]++);
        if (!write_file(path, host.data(), op.bytes)) return ncclSystemError;
    }
    for (const Op& op : ops)
    {
        if (op.send) continue;
        host.resize(op.bytes);
        const std::string path = std::string(dir()) + "/" + op.comm->tag + ".p2p." + std::to_string(op.peer) + "." +
                                 std::to_string(op.comm->rank) + "." + std::to_string(op.comm->recv_seq[op.peer]++);
        if (ncclResult_t r = read_file(path, host.data(), op.bytes); r != ncclSuccess) return r;
        unlink(path.c_str());
        if (op.bytes && hipMemcpy(op.ptr, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t p2p(bool send, void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (!c || peer < 0 || peer >= c->nranks || type_size(type) == 0) return ncclInvalidArgument;
    if (count && !ptr) return ncclInvalidArgument;
    Op op{send, ptr, count * type_size(type), peer, c, stream};
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
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<MockComm*>(comm);
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
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<unsigned char> host(bytes);
    if (bytes && hipMemcpy(host.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string base = std::string(dir()) + "/" + c->tag + ".gather." + std::to_string(c->gather_seq++) + ".";
    if (!write_file(base + std::to_string(c->rank), host.data(), bytes)) return ncclSystemError;
    for (int r = 0; r < c->nranks; r++)
    {
        if (ncclResult_t res = read_file(base + std::to_string(r), host.data(), bytes); res != ncclSuccess) return res;
        if (bytes && hipMemcpy((unsigned char*) recv + (size_t) r * bytes, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    return ncclSuccess;
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
