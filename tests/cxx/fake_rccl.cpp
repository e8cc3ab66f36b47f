// tests/cxx/fake_rccl.cpp -- TEST ONLY.  A stand-in for librccl.so with the seven entry points voldor_amd/csrc/vk_dist.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString, ncclGetVersion), loaded through
// VOLDOR_HIP_RCCL.  Ranks are processes that may SHARE one GPU (RCCL itself refuses two ranks on one device); a collective stages its
// buffers through the host and exchanges them through files in a directory named after the communicator id.  Purpose: run the N > 1
// paths of vk_dist.hip -- record placement, empty steps, uneven shards, barrier / max -- on the one-GPU boxes this project is developed
// on (tests/test_gpu_dist_nccl.py::test_capi_two_ranks_on_one_gpu_through_the_file_backed_stand_in).  What it cannot show is RCCL itself.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    int rank, world;
    long seq;
    std::string dir;
};

static std::string hex_of(const ncclUniqueId& id) {
    char buf[40];
    for (int i = 0; i < 16; i++) snprintf(buf + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    return std::string(buf, 32);
}
static bool read_all(const std::string& p, void* dst, size_t n) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    const size_t got = fread(dst, 1, n, f);
    fclose(f);
    return got == n;
}
static bool write_atomic(const std::string& p, const void* src, size_t n) {
    const std::string tmp = p + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const size_t put = fwrite(src, 1, n, f);
    fclose(f);
    return put == n && rename(tmp.c_str(), p.c_str()) == 0;
}
// every rank publishes `n` bytes, then collects the `n` bytes of every rank (in rank order) into out[world * n]
static ncclResult_t exchange(ncclComm* c, const void* mine, size_t n, std::vector<char>& out) {
    const long s = c->seq++;
    if (!write_atomic(c->dir + "/s" + std::to_string(s) + "_r" + std::to_string(c->rank), mine, n)) return ncclSystemError;
    out.resize(n * (size_t)c->world);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->world; r++) {
        const std::string p = c->dir + "/s" + std::to_string(s) + "_r" + std::to_string(r);
        while (!read_all(p, out.data() + n * (size_t)r, n)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    return ncclSuccess;
}

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int* v) { *v = 999; return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake_rccl: exchange failed"; }
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    const unsigned long long a = (unsigned long long)getpid(), b = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    memcpy(id->internal, &a, 8); memcpy(id->internal + 8, &b, 8);
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    ncclComm* c = new ncclComm{ rank, nranks, 0, std::string("/tmp/fake_rccl_") + hex_of(id) };
    mkdir(c->dir.c_str(), 0700);
    *comm = c;
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    if (type != ncclFloat) return ncclInvalidArgument;
    const size_t n = count * sizeof(float);
    std::vector<char> mine(n), all;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(mine.data(), send, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (ncclResult_t r = exchange(comm, mine.data(), n, all)) return r;
    return hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (type != ncclDouble || op != ncclMax || count != 1) return ncclInvalidArgument;
    double mine = 0.0;
    std::vector<char> all;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(&mine, send, sizeof mine, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (ncclResult_t r = exchange(comm, &mine, sizeof mine, all)) return r;
    double m = mine;
    for (int r = 0; r < comm->world; r++) { double v; memcpy(&v, all.data() + sizeof(double) * r, sizeof v); if (v > m) m = v; }
    return hipMemcpy(recv, &m, sizeof m, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
}
