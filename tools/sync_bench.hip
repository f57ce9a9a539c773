// in-kernel barrier cost on MI355X: persistent kernel, R rounds of {write, release, arrive, spin, acquire, read}
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: one group of all WGs; mode 1: groups of `gsz` WGs sharing blockIdx%8 (same XCD if round-robin)
// mode 2: groups of gsz consecutive WGs (spread over XCDs)
__global__ __launch_bounds__(256) void k(int rounds, unsigned* ctr, float* buf, int nwg, int gsz, int mode, int payload, float* out, int fmode) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  int group, rankInGroup;
  if (mode == 0) { group = 0; rankInGroup = wg; gsz = nwg; }
  else if (mode == 1) { const int x = wg & 7, idx = wg >> 3; group = x * (nwg / 8 / gsz) + idx / gsz; rankInGroup = idx % gsz; }
  else { group = wg / gsz; rankInGroup = wg % gsz; }
  // partner: next WG of the same group
  int partner;
  { const int nr = (rankInGroup + 1) % gsz;
    if (mode == 0) partner = nr; else if (mode == 1) { const int x = wg & 7, base = ((wg >> 3) / gsz) * gsz; partner = ((base + nr) << 3) | x; } else partner = group * gsz + nr; }
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float* b = buf + (size_t)(r & 1) * nwg * 1024;
    if (fmode == 3) { for (int i = 0; i < payload; ++i) __hip_atomic_store(&b[(size_t)wg * 1024 + i * 256 + tid], acc + r + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else for (int i = 0; i < payload; ++i) b[(size_t)wg * 1024 + i * 256 + tid] = acc + r + i;
    if (fmode == 1) __threadfence();
    else if (fmode == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else if (fmode == 3) __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&ctr[group * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gsz;
      while (__hip_atomic_load(&ctr[group * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (fmode == 1) __threadfence();
    else if (fmode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (fmode == 3) { for (int i = 0; i < payload; ++i) acc += __hip_atomic_load(&b[(size_t)partner * 1024 + i * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else for (int i = 0; i < payload; ++i) acc += b[(size_t)partner * 1024 + i * 256 + tid];
  }
  out[wg * 256 + tid] = acc;
}

int main() {
  const int nwg = 256;
  unsigned* ctr; float *buf, *out;
  CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&buf, 2 * nwg * 1024 * 4)); CK(hipMalloc(&out, nwg * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  struct { int mode, gsz, payload; const char* name; } cfgs[] = {
    {0, 256, 0, "global barrier, no payload"}, {0, 256, 1, "global barrier, 1 KB/WG"}, {0, 256, 4, "global barrier, 4 KB/WG"},
    {1, 16, 0, "XCD-local groups of 16, no payload"}, {1, 16, 1, "XCD-local groups of 16, 1 KB"}, {1, 16, 4, "XCD-local groups of 16, 4 KB"},
    {1, 32, 1, "XCD-local groups of 32, 1 KB"},
    {2, 16, 0, "consecutive groups of 16 (spread), no payload"}, {2, 16, 1, "consecutive groups of 16 (spread), 1 KB"},
  };
  for (int fmode = 0; fmode < 4; ++fmode) {
  printf("--- fence mode %d (0 none, 1 __threadfence, 2 amdgcn_fence agent rel/acq, 3 agent-scope atomic ld/st, no fence)\n", fmode);
  for (auto& c : cfgs) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096 * 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, rounds, ctr, buf, nwg, c.gsz, c.mode, c.payload, out, fmode);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-50s %.3f us/round\n", c.name, best * 1e3 / rounds);
  }
  }
  return 0;
}
