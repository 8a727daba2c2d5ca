// Grid-barrier latency probe for gfx950: how much does one device-wide barrier cost inside a
// persistent kernel, as a function of the number of participating workgroups, for
//   mode 0: one shared counter (atomic add + spin on it)
//   mode 1: one arrival flag per workgroup (plain release store; wave 0 polls all flags)
// Every round each block writes a slice and, after the barrier, checks its neighbour's slice.
// build: hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o tools/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr long kSpinCap = 1L << 22;  // never hang the box

__device__ __forceinline__ void barrier_counter(unsigned* ctr, unsigned nb, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    gen += nb;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinCap) break;
    }
  }
  __syncthreads();
}

// flags[b] = number of barriers block b has arrived at
__device__ __forceinline__ void barrier_flags(unsigned* flags, unsigned nb, unsigned b, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x < 64) {
    ++gen;
    if (threadIdx.x == 0)
      __hip_atomic_store(flags + b, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (true) {
      bool ok = true;
      for (unsigned j = threadIdx.x; j < nb; j += 64)
        ok &= __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gen;
      if (__all(ok)) break;
      if (++spins > kSpinCap) break;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default for HIP
  }
  __syncthreads();
}

__global__ void k_barriers(unsigned* sync, int mode, int iters, double* data, int n, double* out,
                           int* bad) {
  const unsigned nb = gridDim.x;
  const unsigned b = blockIdx.x;
  unsigned gen = 0;
  double acc = 0;
  int wrong = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) data[i] = it + i;
    if (mode == 0) barrier_counter(sync, nb, gen); else barrier_flags(sync, nb, b, gen);
    const int nbk = (b + 1) % nb;
    for (int i = nbk * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
      const double v = data[i];
      wrong += (v != (double)(it + i));
      acc += v;
    }
    if (mode == 0) barrier_counter(sync, nb, gen); else barrier_flags(sync, nb, b, gen);
  }
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0) out[b] = acc;
}

int main() {
  unsigned* sync; double *data, *out; int* bad;
  const int n = 50000;
  CK(hipMalloc(&sync, 4096 * 4)); CK(hipMalloc(&data, n * 8)); CK(hipMalloc(&out, 4096 * 8));
  CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 500;
  struct { int grid, threads; } cfg[] = {{8, 1024}, {16, 1024}, {32, 1024}, {48, 1024},
                                         {64, 1024}, {64, 512}, {128, 512}, {196, 256}, {256, 256}};
  for (int mode = 0; mode < 2; ++mode)
    for (auto c : cfg) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(sync, 0, 4096 * 4)); CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(e0));
        k_barriers<<<c.grid, c.threads>>>(sync, mode, iters, data, n, out, bad);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep) printf("mode %d blocks %3d x %4d threads: %.3f us per barrier (incl. work), stale reads %d\n",
                        mode, c.grid, c.threads, ms * 1e3 / (2 * iters), hb);
      }
    }
  return 0;
}
