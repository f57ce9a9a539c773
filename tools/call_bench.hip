// What a caller of hl_step(20) + hl_sync pays around the kernels themselves (the round driver's `bench.py --steps 20
// --warmup 5` protocol): 20 steps of two dependent kernels (272 x 512 threads spinning ~8 us, 354 x 256 threads ~4 us),
// issued as ONE graph / two graphs (16 + 4 steps) / direct launches, completion seen through hipStreamSynchronize or
// through a one-thread kernel that stores a sequence number into pinned host memory which the host polls -- after a
// previous call that has just been waited for, after 200 us and after 20 ms of idling.
// Device time stamps (wall_clock64, 100 MHz) of the first workgroup to start and the last to end are mapped onto the host
// clock (calibration kernel streaming its clock into pinned memory), so every call splits into
//   launch latency (host call -> first workgroup runs) | device time | completion latency (last workgroup done -> host knows).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/call_bench tools/call_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static double nowUs() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Stamps { long long first, last; unsigned seq; };

__device__ __forceinline__ void spinUs(float us) {
  const long long t0 = wall_clock64(); const long long dt = (long long)(us * 100.f);
  while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(1);
}
__global__ void body(Stamps* st, float us, float* data) {
  if (threadIdx.x == 0) atomicMin((unsigned long long*)&st->first, (unsigned long long)wall_clock64());
  float v = data[(blockIdx.x * 64 + threadIdx.x) & 4095];
  if (threadIdx.x == 0) spinUs(us);
  __syncthreads();
  data[(blockIdx.x * 64 + threadIdx.x) & 4095] = v + 1.f;
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&st->last, (unsigned long long)wall_clock64());
}
__global__ void notify(Stamps* st, volatile unsigned* hostFlag) {
  const unsigned s = ++st->seq;
  __hip_atomic_store((unsigned*)hostFlag, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void calib(volatile long long* host, int iters) {
  for (int i = 0; i < iters; ++i) { *host = wall_clock64(); __builtin_amdgcn_s_sleep(8); }
}

int main() {
  const int nA = 272, nB = 354, steps = 20;
  Stamps* st; float* data; CK(hipMalloc(&st, sizeof(Stamps))); CK(hipMalloc(&data, 4096 * 4)); CK(hipMemset(data, 0, 4096 * 4));
  unsigned* flag; long long* hclk;
  CK(hipHostMalloc(&flag, 64, hipHostMallocDefault)); CK(hipHostMalloc(&hclk, 64, hipHostMallocDefault)); *flag = 0; *hclk = 0;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMemset(st, 0, sizeof(Stamps))); CK(hipDeviceSynchronize());
  // clock offset: host_us = dev_ticks / 100 + off   (smallest observed difference = shortest transport)
  double off = 1e300;
  hipLaunchKernelGGL(calib, dim3(1), dim3(1), 0, s, (volatile long long*)hclk, 200000);
  { const double tEnd = nowUs() + 20000; while (nowUs() < tEnd) { const long long v = *(volatile long long*)hclk; const double t = nowUs(); if (v) off = std::min(off, t - v / 100.0); } }
  CK(hipStreamSynchronize(s));
  printf("clock offset %.1f us (host - device)\n", off);

  auto enqueueSteps = [&](int n) {
    for (int j = 0; j < n; ++j) {
      hipLaunchKernelGGL(body, dim3(nA), dim3(512), 0, s, st, 8.f, data);
      hipLaunchKernelGGL(body, dim3(nB), dim3(256), 0, s, st, 4.f, data);
    }
  };
  auto capture = [&](int n, bool withNotify, hipGraphExec_t* ge) -> int {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueueSteps(n);
    if (withNotify) hipLaunchKernelGGL(notify, dim3(1), dim3(1), 0, s, st, (volatile unsigned*)flag);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(ge, g, nullptr, nullptr, 0));
    CK(hipGraphUpload(*ge, s)); CK(hipStreamSynchronize(s));
    return 0;
  };
  hipGraphExec_t g20, g16, g4, g20n, g18;
  if (capture(20, false, &g20) || capture(16, false, &g16) || capture(4, false, &g4) || capture(20, true, &g20n) || capture(18, false, &g18)) return 1;
  {   // first launch of a freshly instantiated (and uploaded) graph exec against its later launches, with a warm device
    for (int trial = 0; trial < 3; ++trial) {
      hipGraphExec_t gf; if (capture(20, false, &gf)) return 1;
      CK(hipGraphLaunch(g4, s)); CK(hipStreamSynchronize(s));
      double ts[4];
      for (int rep = 0; rep < 4; ++rep) { const double t0 = nowUs(); CK(hipGraphLaunch(gf, s)); CK(hipStreamSynchronize(s)); ts[rep] = nowUs() - t0; }
      printf("fresh graph exec of 20 steps: launch 1..4 = %.1f %.1f %.1f %.1f us\n", ts[0], ts[1], ts[2], ts[3]);
      CK(hipGraphExecDestroy(gf));
    }
    // ... and a graph that was launched before, then left alone while 200 other graph launches went by
    hipGraphExec_t gf; if (capture(20, false, &gf)) return 1;
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(gf, s)); CK(hipStreamSynchronize(s)); }
    for (int rep = 0; rep < 200; ++rep) CK(hipGraphLaunch(g4, s));
    CK(hipStreamSynchronize(s));
    double ts[3];
    for (int rep = 0; rep < 3; ++rep) { const double t0 = nowUs(); CK(hipGraphLaunch(gf, s)); CK(hipStreamSynchronize(s)); ts[rep] = nowUs() - t0; }
    printf("graph exec launched before, after 200 launches of another graph: %.1f %.1f %.1f us\n", ts[0], ts[1], ts[2]);
  }
  unsigned expect = 0;
  const char* launchNames[] = {"one graph of 20", "graphs 16 + 4", "40 direct launches", "2 steps direct + graph of 18"};
  const char* syncNames[] = {"hipStreamSynchronize", "notify kernel + poll"};
  const double idles[] = {0, 200, 20000};
  for (double idle : idles) for (int lm = 0; lm < 4; ++lm) for (int sm = 0; sm < 2; ++sm) {
    std::vector<double> tot, lat, dev, fin, enq;
    for (int rep = 0; rep < 25; ++rep) {
      // the previous call, waited for (as the bench's barrier does)
      CK(hipGraphLaunch(g4, s)); CK(hipStreamSynchronize(s));
      if (idle > 0) { const double t = nowUs() + idle; while (nowUs() < t) {} }
      Stamps z{0x7fffffffffffffffLL, 0, expect}; CK(hipMemcpy(st, &z, sizeof(z), hipMemcpyHostToDevice));
      CK(hipStreamSynchronize(s)); CK(hipDeviceSynchronize());
      if (idle > 0) { const double t = nowUs() + idle; while (nowUs() < t) {} }
      const double t0 = nowUs();
      if (lm == 0) { if (sm == 1) CK(hipGraphLaunch(g20n, s)); else CK(hipGraphLaunch(g20, s)); }
      else if (lm == 1) { CK(hipGraphLaunch(g16, s)); CK(hipGraphLaunch(g4, s)); }
      else if (lm == 2) enqueueSteps(steps);
      else { enqueueSteps(2); CK(hipGraphLaunch(g18, s)); }
      if (sm == 1 && lm != 0) hipLaunchKernelGGL(notify, dim3(1), dim3(1), 0, s, st, (volatile unsigned*)flag);
      const double tE = nowUs();
      if (sm == 0) CK(hipStreamSynchronize(s));
      else { ++expect; while (*(volatile unsigned*)flag != expect) {} }
      const double t1 = nowUs();
      CK(hipStreamSynchronize(s));
      Stamps r; CK(hipMemcpy(&r, st, sizeof(r), hipMemcpyDeviceToHost));
      if (rep < 5) continue;
      tot.push_back(t1 - t0); enq.push_back(tE - t0);
      lat.push_back(r.first / 100.0 + off - t0); dev.push_back((r.last - r.first) / 100.0); fin.push_back(t1 - (r.last / 100.0 + off));
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("idle %6.0f us | %-28s | %-22s | total %6.1f us (%.2f per step) = launch %5.1f + device %6.1f + completion %5.1f ; enqueue returns after %5.1f\n",
           idle, launchNames[lm], syncNames[sm], med(tot), med(tot) / steps, med(lat), med(dev), med(fin), med(enq));
    fflush(stdout);
  }
  return 0;
}
