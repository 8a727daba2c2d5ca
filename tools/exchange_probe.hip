// Cross-workgroup exchange latency probe (gfx950): cost of one all-to-all "publish a small
// value, read everybody's" round inside a persistent kernel, for
//   mode 0: counter barrier (release/acquire) + plain loads of the payload
//   mode 1: counter barrier with relaxed atomics only (no fences; shows what the fences cost)
//   mode 2: stamped 16-byte slots {value, stamp} (double-buffered by round parity): one agent-scope 16 B store per workgroup,
//           lanes poll everybody's slot with agent-scope 16 B loads until the stamp matches
// build: hipcc -O3 --offload-arch=gfx950 tools/exchange_probe.hip -o tools/exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr long kSpinCap = 1L << 12;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 ld16_agent(const d2* p) {
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st16_agent(d2* p, d2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <bool FENCED>
__device__ __forceinline__ void barrier_counter(unsigned* ctr, unsigned nb, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    gen += nb;
    if (FENCED) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while ((FENCED ? __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                   : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < gen) {
      if (++spins > kSpinCap) break;
    }
  }
  __syncthreads();
}

__global__ void k_exchange(unsigned* ctr, d2* slots, double* plain, int mode, int iters, double* out,
                           int* bad) {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned gen = 0;
  double acc = 0;
  int wrong = 0;
  for (int it = 1; it <= iters; ++it) {
    const double mine = (double)it * 1000.0 + b;
    double got = 0.0;
    if (mode < 2) {
      if (threadIdx.x == 0) {
        if (mode == 0) plain[b] = mine;
        else __hip_atomic_store(plain + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (mode == 0) barrier_counter<true>(ctr, nb, gen); else barrier_counter<false>(ctr, nb, gen);
      if (lane < (int)nb)
        got = mode == 0 ? plain[lane]
                        : __hip_atomic_load(plain + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // a second barrier so nobody overwrites plain[] before everyone has read it
      if (mode == 0) barrier_counter<true>(ctr, nb, gen); else barrier_counter<false>(ctr, nb, gen);
    } else {
      d2* buf = slots + (it & 1) * 64;  // two buffers: a slot is rewritten only after every
      // reader has published the NEXT round, i.e. finished reading this one
      if (threadIdx.x == 0) st16_agent(buf + b, d2{mine, (double)it});
      if (lane < (int)nb) {
        long spins = 0;
        while (true) {
          const d2 v = ld16_agent(buf + lane);
          if (v.y == (double)it) { got = v.x; break; }
          if (++spins > kSpinCap) break;
        }
      }
      __syncthreads();
    }
    if (lane < (int)nb) wrong += (got != (double)it * 1000.0 + lane);
    acc += got;
  }
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0) out[b] = acc;
}

int main() {
  unsigned* ctr; d2* slots; double *plain, *out; int* bad;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&slots, 2 * 64 * 16)); CK(hipMalloc(&plain, 64 * 8));
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  const int grids[] = {8, 16, 25, 32, 49, 64};
  for (int mode = 0; mode < 3; ++mode)
    for (int g : grids) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(slots, 0, 2 * 64 * 16)); CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(e0));
        k_exchange<<<g, 512>>>(ctr, slots, plain, mode, iters, out, bad);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep) printf("mode %d blocks %2d: %.3f us per exchange round, wrong %d\n", mode, g,
                        ms * 1e3 / iters, hb), fflush(stdout);
      }
    }
  return 0;
}
