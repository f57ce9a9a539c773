// smarties_amd/csrc/per.hip -- the prioritised samplers' distribution (dataSamplingAlgo PERrank / PERerr / PERseq;
// ReplayMemory/Sampling.cpp:101-296): before every minibatch the reference rebuilds a std::discrete_distribution<Uint>
// over all stored transitions (episodes).  libstdc++ normalises it with a SEQUENTIAL double-precision accumulate, divides,
// and builds the cumulative table with a SEQUENTIAL partial_sum; the drawn indices depend on every rounding of those two
// chains, so they are kept sequential here -- one wavefront walks the million values -- and cost what they cost the
// reference: milliseconds per step (the samplers are not selected by any shipped settings file, and the importance weights
// they define are not applied in this version of the reference, Approximator.h:196).  The draws themselves
// (generate_canonical<double, 53> + lower_bound, sort / unique / redraw) run in the sampler kernel (tail_dev.h).
#include "dev_common.h"
#include <hipcub/hipcub.hpp>

namespace hl {

// probabilities in table order: flat index = prefix of the episode + step (PERerr, PERrank keys), or one per episode (PERseq)
__global__ __launch_bounds__(256) void per_probs_kernel(PerArgs a) {
  const float EPS = FLT_EPSILON;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = blockIdx.x * 4 + wave; p < a.nEpisodes; p += gridDim.x * 4) {
    const PosRec rec = a.rp.posRec[p];
    const int e = rec.eidTerm & 0x7fffffff, nd = rec.N - 1;
    if (a.algo == HL_SAMPLE_PERSEQ) {       // Sample_impSeq::prepare (:247-249): (avg squared error + eps)^(1/4) x length
      if (lane == 0) a.prob[p] = sqrtf(sqrtf(a.rp.epAgg[(size_t)e * AGG_N + AGG_AVGSQERR] + EPS)) * (float)(unsigned long long)nd;
      continue;
    }
    for (int j = lane; j < nd; j += 64) {
      const float dq = a.rp.DQ[rec.off + j], d2 = dq * dq;
      if (a.algo == HL_SAMPLE_PERERR) a.prob[rec.prefix + j] = sqrtf(sqrtf(d2 + EPS));       // TSample_impErr::prepare (:192-196)
      else { a.key[rec.prefix + j] = d2; a.idx[rec.prefix + j] = (unsigned)(rec.prefix + j); }   // TSample_impRank: ranked below
    }
  }
}
// TSample_impRank::prepare (:137-146): the i-th largest error gets 1 / sqrt(sqrt(i + 1)) (double square roots of an integer,
// rounded to float), transitions whose error is not positive get 1
__global__ __launch_bounds__(256) void per_rank_kernel(PerArgs a, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float P = a.keySorted[i] > 0 ? (float)(1 / sqrt(sqrt((double)(i + 1)))) : 1.f;
  a.prob[a.idxSorted[i]] = P;
}

__device__ __forceinline__ double readlaneD(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, l), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// discrete_distribution::param_type::_M_initialize: sum = accumulate(p, 0.0); p /= sum; cp = partial_sum(p); cp.back() = 1.
// One wavefront: every lane fetches 16 consecutive values (coalesced), the chain then visits them lane by lane through
// v_readlane -- all lanes carry the same accumulator --, the lane whose value it was keeps the running sum it has to store.
__global__ __launch_bounds__(64) void per_scan_kernel(const float* __restrict__ prob, double* __restrict__ cp, long long n) {
  if (n < 2) return;                                   // (the distribution then always returns 0: nothing to build)
  const int lane = threadIdx.x;
  constexpr int U = 16;
  double s = 0.0;
  for (long long base = 0; base < n; base += 64 * U) {
    double x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const long long e = base + (long long)lane * U + j; x[j] = e < n ? (double)prob[e] : 0.0; }
    for (int l = 0; l < 64; ++l) {
#pragma unroll
      for (int j = 0; j < U; ++j) s += readlaneD(x[j], l);        // (+ 0.0 behind the end: exact)
    }
  }
  double acc = 0.0;
  for (long long base = 0; base < n; base += 64 * U) {
    double q[U], mine[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const long long e = base + (long long)lane * U + j; q[j] = e < n ? (double)prob[e] / s : 0.0; mine[j] = 0.0; }
    for (int l = 0; l < 64; ++l) {
#pragma unroll
      for (int j = 0; j < U; ++j) { acc += readlaneD(q[j], l); if (lane == l) mine[j] = acc; }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) { const long long e = base + (long long)lane * U + j; if (e < n) cp[e] = e == n - 1 ? 1.0 : mine[j]; }
  }
}

size_t per_sort_temp_bytes(long long n) {
  size_t bytes = 0;
  hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, (int)n);
  return bytes;
}
hipError_t launch_per_prepare(const PerArgs& a, long long nTransitions, hipStream_t s) {
  int nb = (a.nEpisodes + 3) / 4; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(per_probs_kernel, dim3(nb), dim3(256), 0, s, a);
  long long n = a.algo == HL_SAMPLE_PERSEQ ? a.nEpisodes : nTransitions;
  if (a.algo == HL_SAMPLE_PERRANK && n > 0) {
    size_t bytes = a.tempBytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(a.temp, bytes, a.key, a.keySorted, a.idx, a.idxSorted, (int)n, 0, 32, s);   // stable
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(per_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n);
  }
  hipLaunchKernelGGL(per_scan_kernel, dim3(1), dim3(64), 0, s, a.prob, a.cp, n);
  return hipGetLastError();
}

}  // namespace hl
