// comm.hip — the one real exchange step of the sharded loop-closure path, under the C ABI.
//
// Reference: the ROS 2 fan-out of `GlobalDescriptors` (cslam/global_descriptor_loop_closure_
// detection.py:196-209) and of `InterRobotMatches` (:223-283).  Here: rank = robot, one RCCL
// communicator over xGMI, all-gathers on a stream of the communicator's own so the exchange of
// step t+1 overlaps the scan of step t.
//
// RCCL is resolved at run time (dlopen of librccl.so.1): libcslam_hip.so keeps loading on a box
// without RCCL, and inside a PyTorch process the loader hands back the copy torch already
// mapped (same SONAME), so there is one RCCL in the process.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; nothing links against librccl at build time

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

using namespace cslam;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle) return CSLAM_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(CSLAM_E_COMM, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define CSLAM_SYM(field, name)                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));      \
    if (!g_rccl.field) return fail(CSLAM_E_COMM, "RCCL symbol %s missing", name);
    CSLAM_SYM(GetUniqueId, "ncclGetUniqueId")
    CSLAM_SYM(CommInitRank, "ncclCommInitRank")
    CSLAM_SYM(CommDestroy, "ncclCommDestroy")
    CSLAM_SYM(AllGather, "ncclAllGather")
    CSLAM_SYM(GroupStart, "ncclGroupStart")
    CSLAM_SYM(GroupEnd, "ncclGroupEnd")
    CSLAM_SYM(GetErrorString, "ncclGetErrorString")
    CSLAM_SYM(GetVersion, "ncclGetVersion")
#undef CSLAM_SYM
    g_rccl.handle = h;
    return CSLAM_OK;
}

#define CSLAM_NCCL_TRY(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess)                                                                \
            return ::cslam::fail(CSLAM_E_COMM, "%s failed: %s (%s:%d)", #expr,                \
                                 g_rccl.GetErrorString(_r), __FILE__, __LINE__);              \
    } while (0)

// rows [b][src_stride] → dense [b_max][dim], rows >= b zeroed; one wave per row
__global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ src, int64_t src_stride,
                                                   int b, int b_max, int dim,
                                                   float* __restrict__ dst) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= b_max) return;
    float* d = dst + w * dim;
    if (w < b) {
        const float* s = src + w * src_stride;
        for (int k = lane; k < dim; k += 64) d[k] = s[k];
    } else {
        for (int k = lane; k < dim; k += 64) d[k] = 0.f;
    }
}

__global__ void k_set_i32(int32_t* p, int32_t v) { *p = v; }

// loopback transport: the caller's block into every other rank's slot, one launch
template <class T>
__global__ __launch_bounds__(256) void k_replicate(const T* __restrict__ send, T* __restrict__ recv,
                                                   int64_t per_rank, int nranks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_rank) return;
    T* dst = recv + (int64_t)blockIdx.y * per_rank;
    if (dst != send) dst[i] = send[i];
}

// measurement aid (CSLAM_LOOPBACK_HOG="workgroups,microseconds"): workgroups that hold a CU each for
// a while, the way a collective's kernel does while it waits for its peers
__global__ __launch_bounds__(256) void k_hog(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

}  // namespace

struct cslam_comm {
    ncclComm_t comm = nullptr;
    cslam_allgather_fn custom = nullptr;  // caller-supplied transport instead of RCCL
    void* custom_user = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;   // the communicator's own stream (non-blocking)
    hipEvent_t ev_in = nullptr;     // caller's stream → comm stream
    hipEvent_t ev_out = nullptr;    // comm stream → caller's stream
    DevBuf pack, count;
    int64_t bytes_gathered = 0;     // payload received so far (all ranks' contributions)
    int64_t n_collectives = 0;
};

extern "C" {

int cslam_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return fail(CSLAM_E_INVALID, "comm_unique_id: null");
    CSLAM_TRY(load_rccl());
    ncclUniqueId id;
    CSLAM_NCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == CSLAM_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return CSLAM_OK;
}

int cslam_comm_init(int rank, int nranks, const uint8_t* unique_id, int device_id,
                    cslam_comm_t** out) {
    if (!out || !unique_id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(CSLAM_E_INVALID, "comm_init: bad args (rank %d of %d)", rank, nranks);
    CSLAM_TRY(load_rccl());
    CSLAM_HIP_TRY(hipSetDevice(device_id));
    cslam_comm* c = new cslam_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = device_id;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(CSLAM_E_COMM, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, nranks,
                    device_id, g_rccl.GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    if (e != hipSuccess) {
        cslam_comm_destroy(c);
        return fail(CSLAM_E_HIP, "comm_init: %s", hipGetErrorString(e));
    }
    *out = c;
    return CSLAM_OK;
}

int cslam_comm_init_custom(int rank, int nranks, int device_id, cslam_allgather_fn allgather,
                           void* user, cslam_comm_t** out) {
    if (!out || !allgather || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(CSLAM_E_INVALID, "comm_init_custom: bad args (rank %d of %d)", rank, nranks);
    CSLAM_HIP_TRY(hipSetDevice(device_id));
    cslam_comm* c = new cslam_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = device_id;
    c->custom = allgather;
    c->custom_user = user;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    if (e != hipSuccess) {
        cslam_comm_destroy(c);
        return fail(CSLAM_E_HIP, "comm_init_custom: %s", hipGetErrorString(e));
    }
    *out = c;
    return CSLAM_OK;
}

namespace {
int loopback_allgather(void* user, const void* send_dev, void* recv_dev, int64_t bytes, void* stream) {
    const int nranks = (int)(intptr_t)user;
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t al = reinterpret_cast<uintptr_t>(send_dev) | reinterpret_cast<uintptr_t>(recv_dev) |
                         (uintptr_t)bytes;
    if ((al & 15) == 0) {
        const int64_t n = bytes / 16;
        hipLaunchKernelGGL((k_replicate<uint4>), dim3((unsigned)((n + 255) / 256), nranks), dim3(256), 0,
                           st, (const uint4*)send_dev, (uint4*)recv_dev, n, nranks);
    } else if ((al & 3) == 0) {
        const int64_t n = bytes / 4;
        hipLaunchKernelGGL((k_replicate<uint32_t>), dim3((unsigned)((n + 255) / 256), nranks), dim3(256),
                           0, st, (const uint32_t*)send_dev, (uint32_t*)recv_dev, n, nranks);
    } else {
        hipLaunchKernelGGL((k_replicate<uint8_t>), dim3((unsigned)((bytes + 255) / 256), nranks),
                           dim3(256), 0, st, (const uint8_t*)send_dev, (uint8_t*)recv_dev, bytes, nranks);
    }
    static const char* hog = getenv("CSLAM_LOOPBACK_HOG");
    if (hog) {
        int wgs = 0, us = 0;
        if (sscanf(hog, "%d,%d", &wgs, &us) == 2 && wgs > 0 && us > 0)
            hipLaunchKernelGGL(k_hog, dim3(wgs), dim3(256), 0, st, (long long)us * 100);  // 100 MHz clock
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace

int cslam_comm_init_loopback(int rank, int nranks, int device_id, cslam_comm_t** out) {
    return cslam_comm_init_custom(rank, nranks, device_id, loopback_allgather,
                                  (void*)(intptr_t)nranks, out);
}

int cslam_comm_destroy(cslam_comm_t* c) {
    if (!c) return CSLAM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);  // custom transports have none
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->pack.release();
    c->count.release();
    delete c;
    return CSLAM_OK;
}

int cslam_comm_rank(const cslam_comm_t* c) { return c ? c->rank : 0; }
int cslam_comm_size(const cslam_comm_t* c) { return c ? c->nranks : 1; }
void* cslam_comm_stream(const cslam_comm_t* c) { return c ? (void*)c->stream : nullptr; }

int cslam_comm_stats(const cslam_comm_t* c, int64_t* bytes_gathered, int64_t* n_collectives,
                     int32_t* rccl_version) {
    if (!c) return fail(CSLAM_E_INVALID, "comm_stats: null");
    if (bytes_gathered) *bytes_gathered = c->bytes_gathered;
    if (n_collectives) *n_collectives = c->n_collectives;
    if (rccl_version) {
        int v = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
        *rccl_version = v;
    }
    return CSLAM_OK;
}

int cslam_comm_allgather(cslam_comm_t* c, const void* send_dev, void* recv_dev,
                         int64_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev || bytes_per_rank < 0)
        return fail(CSLAM_E_INVALID, "comm_allgather: bad args");
    if (bytes_per_rank == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream == CSLAM_COMM_OWN_STREAM ? c->stream : (hipStream_t)stream;
    if (c->custom) {
        const int rc = c->custom(c->custom_user, send_dev, recv_dev, bytes_per_rank, (void*)st);
        if (rc != 0) return fail(CSLAM_E_COMM, "custom all-gather transport returned %d", rc);
    } else {
        CSLAM_NCCL_TRY(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclChar, c->comm, st));
    }
    c->bytes_gathered += bytes_per_rank * c->nranks;
    c->n_collectives += 1;
    return CSLAM_OK;
}

int cslam_allgather_descriptors(cslam_comm_t* c, const float* local_dev, int64_t row_stride,
                                int32_t b, int32_t b_max, int32_t dim, float* all_dev,
                                int32_t* counts_dev, void* stream) {
    if (!c || b < 0 || b_max <= 0 || b > b_max || dim <= 0 || !all_dev || !counts_dev ||
        (b > 0 && (!local_dev || row_stride < dim)))
        return fail(CSLAM_E_INVALID, "allgather_descriptors: bad args");
    CSLAM_HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream == CSLAM_COMM_OWN_STREAM ? c->stream : (hipStream_t)stream;
    const size_t blk = (size_t)b_max * dim * sizeof(float);
    // this rank's block goes straight into its slot of the gathered array (in-place all-gather)
    float* mine = all_dev + (size_t)c->rank * b_max * dim;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((b_max + 3) / 4)), dim3(256), 0, st, local_dev,
                       row_stride, b, b_max, dim, mine);
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, st, counts_dev + c->rank, b);
    CSLAM_HIP_TRY(hipGetLastError());
    if (c->custom) {
        CSLAM_TRY(cslam_comm_allgather(c, mine, all_dev, (int64_t)blk, (void*)st));
        return cslam_comm_allgather(c, counts_dev + c->rank, counts_dev, sizeof(int32_t), (void*)st);
    }
    CSLAM_NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.AllGather(mine, all_dev, blk, ncclChar, c->comm, st);
    ncclResult_t r2 = g_rccl.AllGather(counts_dev + c->rank, counts_dev, sizeof(int32_t), ncclChar,
                                       c->comm, st);
    CSLAM_NCCL_TRY(g_rccl.GroupEnd());
    CSLAM_NCCL_TRY(r1);
    CSLAM_NCCL_TRY(r2);
    c->bytes_gathered += (int64_t)(blk + sizeof(int32_t)) * c->nranks;
    c->n_collectives += 2;
    return CSLAM_OK;
}

int cslam_device_copy(void* dst, const void* src, int64_t bytes, int kind, void* stream, int sync) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src)) || kind < 0 || kind > 2)
        return fail(CSLAM_E_INVALID, "device_copy: bad args");
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice
                                      : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    hipStream_t st = (hipStream_t)stream;
    if (bytes > 0) CSLAM_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, k, st));
    if (sync) CSLAM_HIP_TRY(hipStreamSynchronize(st));
    return CSLAM_OK;
}

}  // extern "C"
