// voldor_amd/csrc/vk_dist.hip -- the multi-GPU exchange of the VO hot path, below the C-ABI.
//
// The path shards across independent sequences only: one process per GPU, rank g runs the windows of sequence g on its own
// device Context (vk_voldor_device_block), and after every batch step each rank needs the pose block of every sequence --
// ONE ncclAllGather of 1 + 42 N floats per rank (211 floats = 844 B for N = 5) over RCCL / xGMI.  No other collective exists on
// this path (the reference has no multi-GPU path at all: file-static device buffers, default stream).  Host code is C++: the
// communicator, its stream and the send / receive records live here; a launcher (C++, torchrun + ctypes, MPI ...) only has to
// carry the 128-byte ncclUniqueId from rank 0 to the others -- or point every rank at one file (vk_dist_init_file).
//
// RCCL is bound at the first vk_dist_* call (dlopen of librccl.so.1), not at load time: single-GPU users of libvoldor_hip.so need
// the HIP runtime only, and a process that already carries an RCCL (torch) is not forced to map a second one at import.
#include "vk_common.hpp"
#include "vk_internal.hpp"
#include "../../include/voldor_hip.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <algorithm>
#include <vector>

namespace vk {
namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
struct Dist {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 0, device = -1;
    float* send = nullptr; float* recv = nullptr; size_t cap = 0;  // device records (floats per rank: cap)
    double* scalar = nullptr;                                      // device scalar for barrier / max
    hipEvent_t ev0 = nullptr, ev1 = nullptr;                       // around the ncclAllGather on `stream` (vk_dist_allgather_stats)
    std::vector<float> ag_dev_us, ag_host_us;                      // per all-gather: HIP-event time on the stream / host wall time incl. the wait
};
std::mutex g_dmu;
Rccl g_rccl;
Dist g_dist;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = { getenv("VOLDOR_HIP_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* h = nullptr;
    for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
    if (!h) { fprintf(stderr, "voldor_hip: cannot load RCCL (librccl.so.1): %s\n", dlerror()); return (int)hipErrorSharedObjectInitFailed; }
#define VK_SYM(field, name) do { g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
        if (!g_rccl.field) { fprintf(stderr, "voldor_hip: RCCL lacks %s\n", name); dlclose(h); return (int)hipErrorSharedObjectSymbolNotFound; } } while (0)
    VK_SYM(GetUniqueId, "ncclGetUniqueId"); VK_SYM(CommInitRank, "ncclCommInitRank"); VK_SYM(CommDestroy, "ncclCommDestroy");
    VK_SYM(AllGather, "ncclAllGather"); VK_SYM(AllReduce, "ncclAllReduce"); VK_SYM(GetErrorString, "ncclGetErrorString");
    VK_SYM(GetVersion, "ncclGetVersion");
#undef VK_SYM
    g_rccl.h = h;
    return 0;
}
#define VK_NCCL(expr) do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
        fprintf(stderr, "voldor_hip: %s failed: %s (%s:%d)\n", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); return 1000 + (int)r_; } } while (0)

int ensure_records(size_t floats_per_rank) {
    Dist& d = g_dist;
    if (d.cap >= floats_per_rank && d.send) return 0;
    if (d.send) (void)hipFree(d.send);
    if (d.recv) (void)hipFree(d.recv);
    d.send = d.recv = nullptr; d.cap = 0;
    VK_CHECK(hipMalloc((void**)&d.send, sizeof(float) * floats_per_rank));
    VK_CHECK(hipMalloc((void**)&d.recv, sizeof(float) * floats_per_rank * (size_t)d.world));
    d.cap = floats_per_rank;
    return 0;
}
int allgather_locked(const float* send_dev, float* recv_dev, int count) {
    Dist& d = g_dist;
    if (!d.comm) return (int)hipErrorNotInitialized;
    // BASELINE.md cfg4 asks for the latency of the collective on its own: HIP events around it on the communicator's stream (device
    // time of the exchange) and the host's wall clock from issue to completion (what a step pays for it)
    const auto t0 = std::chrono::steady_clock::now();
    const bool timed = d.ev0 && d.ev1 && hipEventRecord(d.ev0, d.stream) == hipSuccess;  // timing is optional: an event that cannot be recorded must not keep this rank out of the collective (the peers are waiting in it)
    VK_NCCL(g_rccl.AllGather(send_dev, recv_dev, (size_t)count, ncclFloat, d.comm, d.stream));
    const bool timed1 = timed && hipEventRecord(d.ev1, d.stream) == hipSuccess;
    VK_CHECK(hipStreamSynchronize(d.stream));
    if (timed1 && d.ag_dev_us.size() < (size_t)VK_DIST_STATS_MAX) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, d.ev0, d.ev1) == hipSuccess) {
            d.ag_dev_us.push_back(ms * 1e3f);
            d.ag_host_us.push_back((float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    return 0;
}
__global__ void k_mark_empty(float* blk, int n, float tag, float code) {  // "no sequence in this slot": n_registered = -1 (or VK_DIST_FAILED + the error code), the rest zero
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) blk[i] = i == 0 ? tag : (i == 1 ? code : 0.f);
}
}  // namespace
}  // namespace vk

using namespace vk;

extern "C" {
int vk_dist_get_unique_id(void* id_out) {
    std::lock_guard<std::mutex> lk(g_dmu);
    if (!id_out) return (int)hipErrorInvalidValue;
    if (int e = load_rccl()) return e;
    static_assert(sizeof(ncclUniqueId) == VK_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    VK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int vk_dist_init(int rank, int world, const void* id_in) {
    std::lock_guard<std::mutex> lk(g_dmu);
    if (!id_in || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
    if (g_dist.comm) { fprintf(stderr, "voldor_hip: vk_dist_init called twice (vk_dist_finalize first)\n"); return (int)hipErrorInvalidValue; }
    if (int e = load_rccl()) return e;
    Dist& d = g_dist;
    VK_CHECK(hipGetDevice(&d.device));  // the device the caller selected (vk_set_device / hipSetDevice): one process per GPU
    VK_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    const ncclResult_t r = g_rccl.CommInitRank(&d.comm, world, id, rank);
    if (r != ncclSuccess || hipMalloc((void**)&d.scalar, sizeof(double)) != hipSuccess || hipEventCreate(&d.ev0) != hipSuccess ||
        hipEventCreate(&d.ev1) != hipSuccess) {  // leave nothing behind: a later vk_dist_init starts clean
        fprintf(stderr, "voldor_hip: ncclCommInitRank(rank %d of %d) failed: %s\n", rank, world, r != ncclSuccess ? g_rccl.GetErrorString(r) : "out of device memory");
        if (d.comm) g_rccl.CommDestroy(d.comm);
        if (d.scalar) (void)hipFree(d.scalar);
        if (d.ev0) (void)hipEventDestroy(d.ev0);
        if (d.ev1) (void)hipEventDestroy(d.ev1);
        (void)hipStreamDestroy(d.stream);
        d = Dist{};
        return r != ncclSuccess ? 1000 + (int)r : (int)hipErrorOutOfMemory;
    }
    d.rank = rank; d.world = world;
    // the exchange buffers for the largest record there is (MAX_FRAMES frames: 2.7 KB per rank) now, so that a batch step has nothing left that
    // could fail between "my window is done" and the collective the peers are already waiting in
    if (int e = ensure_records((size_t)(1 + 42 * MAX_FRAMES))) return e;
    return 0;
}

/* Rendezvous through one file every rank can see (a launcher without a store of its own): rank 0 creates the id and publishes it
 * by an atomic rename; the others wait for the file.  The path must be NEW for every job (as with any file rendezvous: a file left by
 * an earlier job would hand its stale id to ranks that start before rank 0 has replaced it); rank 0 removes a leftover it finds, which
 * covers jobs run one after the other under one name; with a job tag in the environment (VOLDOR_HIP_JOB_ID, or torchrun's
 * TORCHELASTIC_RUN_ID) a leftover of ANOTHER job is recognised and ignored by the waiting ranks as well.  The file is left in place
 * (the launcher owns the path). */
int vk_dist_init_file(int rank, int world, const char* path, int timeout_s) {
    if (!path || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
    // File = 64-byte job tag + the id.  The tag is what the launcher says identifies THIS job (VOLDOR_HIP_JOB_ID, else torchrun's
    // TORCHELASTIC_RUN_ID, else empty): a rank that starts before rank 0 has replaced a file left by an earlier job under the same
    // path reads that job's tag, sees it is not its own and keeps waiting instead of joining a communicator nobody else is in.
    char tag[64];
    memset(tag, 0, sizeof tag);
    const char* job = getenv("VOLDOR_HIP_JOB_ID");
    if (!job || !*job) job = getenv("TORCHELASTIC_RUN_ID");
    if (job) strncpy(tag, job, sizeof tag - 1);
    unsigned char id[VK_DIST_ID_BYTES];
    if (rank == 0) {
        (void)remove(path);
        if (int e = vk_dist_get_unique_id(id)) return e;
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f) return (int)hipErrorFileNotFound;
        const size_t n = fwrite(tag, 1, sizeof tag, f) + fwrite(id, 1, sizeof id, f);
        fclose(f);
        if (n != sizeof tag + sizeof id || rename(tmp.c_str(), path) != 0) return (int)hipErrorFileNotFound;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                char ftag[64];
                const size_t n = fread(ftag, 1, sizeof ftag, f) + fread(id, 1, sizeof id, f);
                fclose(f);
                if (n == sizeof ftag + sizeof id && memcmp(ftag, tag, sizeof tag) == 0) break;  // (another job's file: not ours, wait for rank 0)
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s > 0 ? timeout_s : 120)) {
                fprintf(stderr, "voldor_hip: rank %d timed out waiting for %s\n", rank, path);
                return (int)hipErrorNotReady;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return vk_dist_init(rank, world, id);
}

int vk_dist_rank(void) { std::lock_guard<std::mutex> lk(g_dmu); return g_dist.comm ? g_dist.rank : -1; }
int vk_dist_world(void) { std::lock_guard<std::mutex> lk(g_dmu); return g_dist.comm ? g_dist.world : 0; }
int vk_dist_rccl_version(void) { std::lock_guard<std::mutex> lk(g_dmu); int v = 0; if (load_rccl() == 0) g_rccl.GetVersion(&v); return v; }

int vk_dist_allgather(const float* send_dev, float* recv_dev, int count) {
    std::lock_guard<std::mutex> lk(g_dmu);
    if (!send_dev || !recv_dev || count <= 0) return (int)hipErrorInvalidValue;
    return allgather_locked(send_dev, recv_dev, count);
}

/* max over the ranks of a host double (the bench's step time); also a barrier */
int vk_dist_allreduce_max(double* io_host) {
    std::lock_guard<std::mutex> lk(g_dmu);
    Dist& d = g_dist;
    if (!d.comm) return (int)hipErrorNotInitialized;
    if (!io_host) return (int)hipErrorInvalidValue;
    VK_CHECK(hipMemcpyAsync(d.scalar, io_host, sizeof(double), hipMemcpyHostToDevice, d.stream));
    VK_NCCL(g_rccl.AllReduce(d.scalar, d.scalar, 1, ncclDouble, ncclMax, d.comm, d.stream));
    VK_CHECK(hipMemcpyAsync(io_host, d.scalar, sizeof(double), hipMemcpyDeviceToHost, d.stream));
    VK_CHECK(hipStreamSynchronize(d.stream));
    return 0;
}
int vk_dist_barrier(void) { double z = 0.0; return vk_dist_allreduce_max(&z); }

/* One batch step of the sharded job: this rank's window (flows == NULL: this rank has no sequence in this step) through
 * vk_voldor_device_block, then the all-gather.  all_blocks_host[world][1 + 42 N]: rank r's record { n_registered | poses[N][6] |
 * poses_covar[N][36] }, n_registered = -1 for an empty slot.  The remaining arguments are those of vk_voldor_device. */
int vk_voldor_sharded(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                      const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                      float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                      float* poses_covar, float* depth, float* depth_conf, float* all_blocks_host) {
    if (N < 1 || N > MAX_FRAMES || !all_blocks_host) return (int)hipErrorInvalidValue;
    const int len = 1 + 42 * N;
    float* send; float* recv; hipStream_t st; int world;
    {
        std::lock_guard<std::mutex> lk(g_dmu);
        if (!g_dist.comm) return (int)hipErrorNotInitialized;
        if (int e = ensure_records((size_t)len)) return e;  // (cannot fail: vk_dist_init allocated the records for MAX_FRAMES -- nothing between here and the collective returns early)
        send = g_dist.send; recv = g_dist.recv; st = g_dist.stream; world = g_dist.world;
    }
    // A rank whose own window fails must STILL take part in the collective: the other ranks are already inside ncclAllGather +
    // hipStreamSynchronize, which has no timeout -- returning here would hang the whole job on one rank's local error (bad config,
    // out of memory, a lost device).  The failing rank sends the record { VK_DIST_FAILED | error code | 0 ... }, every rank sees which
    // peer failed (n_registered slot == VK_DIST_FAILED, slot 1 = that rank's error code), and this call returns the saved error.
    int local_err = 0;
    if (flows) {
        local_err = vk_voldor_device_block(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy, cx, cy,
                                           basefocal, N, N_dp, w, h, config, n_registered, poses, poses_covar, depth, depth_conf, send);
        // (returns after the window's stream has drained: on success the record is complete)
    } else {
        hipLaunchKernelGGL(k_mark_empty, dim3((len + 255) / 256), dim3(256), 0, st, send, len, -1.f, 0.f);
        local_err = (int)hipGetLastError();
        if (n_registered) *n_registered = -1;
    }
    if (local_err) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();  // a failed window may have left kernels in flight on ITS stream that still write `send`: the marker goes in after them
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_mark_empty, dim3((len + 255) / 256), dim3(256), 0, st, send, len, (float)VK_DIST_FAILED, (float)local_err);
        if (hipGetLastError() != hipSuccess) {  // not even a marker launch: the device is gone.  Send what the buffer holds, marked from the host if that still works
            const float mark[2] = { (float)VK_DIST_FAILED, (float)local_err };
            (void)hipMemcpyAsync(send, mark, sizeof mark, hipMemcpyHostToDevice, st);
        }
        if (n_registered) *n_registered = VK_DIST_FAILED;
    }
    {
        std::lock_guard<std::mutex> lk(g_dmu);
        if (int e = allgather_locked(send, recv, len)) return local_err ? local_err : e;
        VK_CHECK(hipMemcpyAsync(all_blocks_host, recv, sizeof(float) * (size_t)len * world, hipMemcpyDeviceToHost, st));
        VK_CHECK(hipStreamSynchronize(st));
    }
    return local_err;
}

/* Latency of the all-gathers issued so far (vk_voldor_sharded, vk_dist_allgather): up to `cap` samples each of the HIP-event time of the
 * collective on the communicator's stream and of the host's wall clock from issue to completion, microseconds, in call order.
 * *n = samples written.  reset != 0 clears the record afterwards. */
int vk_dist_allgather_stats(float* dev_us, float* host_us, int cap, int* n, int reset) {
    std::lock_guard<std::mutex> lk(g_dmu);
    Dist& d = g_dist;
    if (!n || cap < 0) return (int)hipErrorInvalidValue;
    const int m = (int)std::min<size_t>((size_t)cap, d.ag_dev_us.size());
    for (int i = 0; i < m; i++) { if (dev_us) dev_us[i] = d.ag_dev_us[i]; if (host_us) host_us[i] = d.ag_host_us[i]; }
    *n = m;
    if (reset) { d.ag_dev_us.clear(); d.ag_host_us.clear(); }
    return 0;
}

int vk_dist_finalize(void) {
    std::lock_guard<std::mutex> lk(g_dmu);
    Dist& d = g_dist;
    if (d.comm) { (void)hipStreamSynchronize(d.stream); g_rccl.CommDestroy(d.comm); }
    if (d.send) (void)hipFree(d.send);
    if (d.recv) (void)hipFree(d.recv);
    if (d.scalar) (void)hipFree(d.scalar);
    if (d.ev0) (void)hipEventDestroy(d.ev0);
    if (d.ev1) (void)hipEventDestroy(d.ev1);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d = Dist{};
    return 0;
}
}  // extern "C"
