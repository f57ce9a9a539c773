// two-level grid barrier cost: per-group counters (16 WGs each), last arriver bumps a global counter, everyone polls it
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void k(int rounds, unsigned* ctr, unsigned* gctr, int nwg, int gsz, int sleepN, float* out) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int x = wg & 7, idx = wg >> 3; const int group = x * (nwg / 8 / gsz) + idx / gsz;
  const int nGroups = nwg / gsz;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&ctr[group * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old % gsz == (unsigned)gsz - 1) __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * nGroups;
      int spins = 0;
      while ((int)(__hip_atomic_load(gctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < 100000) { if (sleepN == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4); }
    }
    __syncthreads();
    acc += 1.f;
  }
  out[wg * 512 + tid] = acc;
}
int main() {
  unsigned *ctr, *gctr; float* out;
  CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&gctr, 256)); CK(hipMalloc(&out, 512 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  for (int nwg : {256, 384, 512}) for (int sleepN : {1, 4}) {
    const int n = nwg / 16 * 16;   // nwg must be a multiple of 128 for the same-XCD mapping; others just approximate
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096 * 4)); CK(hipMemset(gctr, 0, 256)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(n), dim3(512), 0, 0, rounds, ctr, gctr, n, 16, sleepN, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("two-level barrier, %d WGs of 512 threads, sleep %d: %.3f us/round\n", n, sleepN, best * 1e3 / rounds);
  }
  return 0;
}
