// Two kernel chains on two streams, coupled only by device counters (no cross-stream edges):
//   stream A: K1(0), K1(1), ...   K1(j) waits until all workgroups of K2(j-1) have arrived
//   stream B: K2(0), K2(1), ...   K2(j) waits until all workgroups of K1(j) have arrived
// Each kernel: prologue (PRE us of clock spinning = work that needs nothing from the other chain), wait,
// WORK us of spinning, publish (agent-scope store + counter add).  Compared with the same kernels in ONE
// stream without waits (every kernel does PRE + WORK).  Everything replays from hipGraphs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Ctl { unsigned doneA[8 * 32]; unsigned doneB[8 * 32]; int err; };   // 8 counters per chain, one 128-byte line each

__device__ __forceinline__ void spinUs(float us) {   // wall_clock64: 100 MHz
  const long long t0 = wall_clock64(); const long long dt = (long long)(us * 100.f);
  while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(1);
}
// waits until the 8 counters at c sum to >= target
__device__ __forceinline__ bool waitSum(const unsigned* c, unsigned target, int* err) {
  int spins = 0;
  for (;;) {
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += __hip_atomic_load(c + i * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(s - target) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 16)) { *err = 1; return false; }
  }
}
template <int WHICH>   // 0 = K1 (chain A), 1 = K2 (chain B)
__global__ __launch_bounds__(512) void kern(Ctl* ctl, int nwgA, int nwgB, float pre, float work, int coupled, float* data, float* out) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  unsigned* mine = WHICH == 0 ? ctl->doneA : ctl->doneB;
  const unsigned* other = WHICH == 0 ? ctl->doneB : ctl->doneA;
  const int nMine = WHICH == 0 ? nwgA : nwgB, nOther = WHICH == 0 ? nwgB : nwgA;
  __shared__ unsigned sJ;
  if (tid == 0) {
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += __hip_atomic_load(mine + i * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sJ = s / nMine;          // my launch index: all workgroups of my predecessor in this stream have arrived
  }
  __syncthreads();
  const unsigned j = sJ;
  if (tid == 0) spinUs(pre);
  __syncthreads();
  if (coupled) {
    if (tid == 0) { const unsigned need = WHICH == 0 ? j * nOther : (j + 1) * nOther; if (need) waitSum(other, need, &ctl->err); }
    __syncthreads();
  }
  // consume something the other chain published (agent-scope load), do the work, publish
  float v = __hip_atomic_load(data + (WHICH ? 0 : 4096) + ((wg * 64 + tid) & 4095), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) spinUs(work);
  __syncthreads();
  __hip_atomic_store(data + (WHICH ? 4096 : 0) + ((wg * 64 + tid) & 4095), v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(mine + (wg & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0 && wg == 0) out[0] = v;
}

int main(int argc, char** argv) {
  const int nA = 272, nB = 354, steps = 500;
  Ctl* ctl; float *data, *out;
  CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&data, 8192 * 4)); CK(hipMalloc(&out, 1024));
  hipStream_t sA, sB; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  struct Cfg { float preA, workA, preB, workB; };
  const Cfg cfgs[] = {{0.f, 0.f, 0.f, 0.f}, {3.f, 8.f, 2.f, 3.f}, {0.f, 8.f, 0.f, 3.f}, {3.f, 0.f, 2.f, 0.f}};
  for (const Cfg& c : cfgs) for (int mode = 0; mode < 3; ++mode) {
    // mode 0: one stream, uncoupled (today's structure)   mode 1: two streams coupled, graph   mode 2: two streams coupled, eager
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(data, 0, 8192 * 4)); CK(hipDeviceSynchronize());
    auto enqueue = [&]() {
      for (int j = 0; j < steps; ++j) {
        if (mode == 0) {
          hipLaunchKernelGGL(kern<0>, dim3(nA), dim3(512), 0, sA, ctl, nA, nB, c.preA, c.workA, 0, data, out);
          hipLaunchKernelGGL(kern<1>, dim3(nB), dim3(512), 0, sA, ctl, nA, nB, c.preB, c.workB, 0, data, out);
        } else {
          hipLaunchKernelGGL(kern<0>, dim3(nA), dim3(512), 0, sA, ctl, nA, nB, c.preA, c.workA, 1, data, out);
          hipLaunchKernelGGL(kern<1>, dim3(nB), dim3(512), 0, sB, ctl, nA, nB, c.preB, c.workB, 1, data, out);
        }
      }
    };
    float ms = 0;
    if (mode != 2) {
      CK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
      if (mode == 1) { CK(hipEventRecord(fork, sA)); CK(hipStreamWaitEvent(sB, fork, 0)); }
      enqueue();
      if (mode == 1) { CK(hipEventRecord(join, sB)); CK(hipStreamWaitEvent(sA, join, 0)); }
      CK(hipStreamEndCapture(sA, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sA)); CK(hipGraphLaunch(ge, sA)); CK(hipEventRecord(e1, sA)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < ms) ms = t;
      }
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    } else {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sA)); CK(hipEventRecord(fork, sA)); CK(hipStreamWaitEvent(sB, fork, 0));
        enqueue();
        CK(hipEventRecord(join, sB)); CK(hipStreamWaitEvent(sA, join, 0));
        CK(hipEventRecord(e1, sA)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < ms) ms = t;
      }
    }
    int err; CK(hipMemcpy(&err, &ctl->err, 4, hipMemcpyDeviceToHost));
    printf("pre/work A %.0f/%.0f B %.0f/%.0f  mode %d (%s): %.2f us per step  err %d\n", c.preA, c.workA, c.preB, c.workB, mode,
           mode == 0 ? "one stream, graph" : mode == 1 ? "two coupled streams, graph" : "two coupled streams, eager", ms * 1e3 / steps, err);
    fflush(stdout);
  }
  return 0;
}
