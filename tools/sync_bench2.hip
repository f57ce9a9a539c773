// validate + time the in-kernel exchange primitive: sc1 (agent-scope) vector stores/loads + counter barrier, no fences
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_sc1(float* p, float4 v) {
  f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 ld4_sc1(const float* p) {
  f32x4 x; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory"); return make_float4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// groups of gsz consecutive WGs (mode 2 = spread over XCDs) or same-XCD (mode 1). Every WG writes f4PerThread float4/thread,
// then reads the whole group's data (gsz * that) like the fused kernel would (panel exchange).
__global__ __launch_bounds__(256) void k(int rounds, unsigned* ctr, float* buf, int nwg, int gsz, int mode, int f4w, int plain, unsigned* errs) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  int group, rank;
  if (mode == 1) { const int x = wg & 7, idx = wg >> 3; group = x * (nwg / 8 / gsz) + idx / gsz; rank = idx % gsz; }
  else { group = wg / gsz; rank = wg % gsz; }
  const size_t wgFloats = (size_t)f4w * 256 * 4;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* b = buf + (size_t)(r & 1) * nwg * wgFloats + (size_t)group * gsz * wgFloats;
    for (int i = 0; i < f4w; ++i) {
      const float v = (float)(r * 7 + rank * 3 + i);
      float* p = b + (size_t)rank * wgFloats + ((size_t)i * 256 + tid) * 4;
      if (plain == 1) *reinterpret_cast<float4*>(p) = make_float4(v, v + 1, v + 2, (float)tid);
      else st4_sc1(p, make_float4(v, v + 1, v + 2, (float)tid));
    }
    wait_vm0();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&ctr[group * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gsz;
      while (__hip_atomic_load(&ctr[group * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // read the whole group's payload: gsz * f4w float4 per thread... too much for big gsz; read own slice from each member
    for (int i = 0; i < f4w; i += 4) {   // a quarter of each member's payload, all loads in flight at once
      float4 v[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float* p = b + (size_t)m * wgFloats + ((size_t)i * 256 + tid) * 4;
        if (plain) v[m] = *reinterpret_cast<const float4*>(p); else v[m] = ld4_sc1(p);
      }
      if (!plain) wait_vm0();
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float e = (float)(r * 7 + m * 3 + i);
        if (v[m].x != e || v[m].y != e + 1 || v[m].z != e + 2 || v[m].w != (float)tid) ++bad;
      }
    }
    // second barrier so nobody overwrites buffer (r&1) two rounds later before all have read: double buffering + this is enough
  }
  if (bad) atomicAdd(errs, bad);
}

int main() {
  const int nwg = 256;
  unsigned *ctr, *errs; float* buf;
  CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&buf, (size_t)2 * nwg * 4 * 256 * 4 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  struct { int mode, gsz, f4w, plain; const char* name; } cfgs[] = {
    {1, 16, 1, 0, "same-XCD groups of 16, sc1, 4 KB/WG"}, {1, 16, 4, 0, "same-XCD groups of 16, sc1, 16 KB/WG"},
    {2, 16, 1, 0, "spread groups of 16, sc1, 4 KB/WG"}, {2, 16, 4, 0, "spread groups of 16, sc1, 16 KB/WG"},
    {1, 16, 1, 1, "same-XCD groups of 16, PLAIN ld/st, 4 KB/WG"}, {2, 16, 1, 1, "spread groups of 16, PLAIN ld/st, 4 KB/WG"},
    {1, 16, 1, 2, "same-XCD groups of 16, sc1 st + plain ld, 4 KB/WG"}, {2, 16, 1, 2, "spread groups of 16, sc1 st + plain ld, 4 KB/WG"},
  };
  for (auto& c : cfgs) {
    float best = 1e9; unsigned herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096 * 4)); CK(hipMemset(errs, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, rounds, ctr, buf, nwg, c.gsz, c.mode, c.f4w, c.plain, errs);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      unsigned e; CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); herr += e;
    }
    printf("%-50s %.3f us/round   mismatches %u\n", c.name, best * 1e3 / rounds, herr);
  }
  return 0;
}
