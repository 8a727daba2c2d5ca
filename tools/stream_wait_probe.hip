// Does a kernel-written flag + hipStreamWaitValue32 order a second stream behind a kernel of the first - and what
// does it cost the FIRST stream against hipEventRecord + hipStreamWaitEvent?  (round 4: the sharded step records an
// event behind its record kernel for the record-exchange stream; the packet sits between two scans.)
// build: hipcc --offload-arch=gfx950 -O2 -o stream_wait_probe tools/stream_wait_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(float* data, int n, int rounds, uint32_t* flag, uint32_t value, uint32_t* ticket) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? data[i] : 0.f;
    for (int r = 0; r < rounds; ++r) v = v * 1.0000001f + 0.5f;
    if (i < n) data[i] = v;
    if (flag == nullptr) return;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_consume(const float* data, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = data[i];
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const int n = 1 << 20, steps = 300;
    float *data, *out;
    uint32_t *flag_sig = nullptr, *flag_dev = nullptr, *ticket;
    CK(hipMalloc(&data, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4));
    CK(hipMemset(data, 0, n * 4));
    CK(hipMalloc(&flag_dev, 4)); CK(hipMemset(flag_dev, 0, 4));
    hipError_t es = hipExtMallocWithFlags((void**)&flag_sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal memory): %s\n", hipGetErrorString(es));
    if (es == hipSuccess) CK(hipMemset(flag_sig, 0, 8));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<float> h(n);
    for (int mode = 0; mode < 4; ++mode) {
        // 0: stream a alone; 1: event + wait event; 2: kernel flag in signal memory + wait value; 3: ... in plain device memory
        uint32_t* flag = mode == 2 ? flag_sig : (mode == 3 ? flag_dev : nullptr);
        if (mode == 2 && es != hipSuccess) continue;
        CK(hipMemset(data, 0, n * 4)); CK(hipDeviceSynchronize());
        hipError_t werr = hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) {
            const uint32_t value = 1000u * (mode + 1) + s + 1;
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, a, data, n, 200, flag, value, ticket);
            if (mode == 1) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
            if (mode >= 2) { werr = hipStreamWaitValue32(b, flag, value, hipStreamWaitValueGte, 0xFFFFFFFFu); if (werr != hipSuccess) break; }
            if (mode >= 1) hipLaunchKernelGGL(k_consume, dim3(n / 256), dim3(256), 0, b, data, n, out);
        }
        if (werr != hipSuccess) { printf("mode %d: hipStreamWaitValue32 -> %s\n", mode, hipGetErrorString(werr)); (void)hipDeviceSynchronize(); continue; }
        CK(hipStreamSynchronize(a));
        const double ta = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
        std::vector<float> hd(n); CK(hipMemcpy(hd.data(), data, n * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < n; i += 997) bad += mode >= 1 && h[i] != hd[i];
        printf("mode %d (%s): %.2f us per step on the producing stream, consumer saw the last step's data: %s\n", mode,
               mode == 0 ? "producer alone" : mode == 1 ? "event record + wait event" : mode == 2 ? "kernel flag (signal memory) + wait value" : "kernel flag (hipMalloc) + wait value",
               ta, bad ? "NO" : "yes");
    }
    return 0;
}
