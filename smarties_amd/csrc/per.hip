// smarties_amd/csrc/per.hip -- the prioritised samplers' distribution (dataSamplingAlgo PERrank / PERerr / PERseq;
// ReplayMemory/Sampling.cpp:101-296): before every minibatch the reference rebuilds a std::discrete_distribution<Uint>
// over all stored transitions (episodes).  libstdc++ normalises it with a SEQUENTIAL double-precision accumulate, divides,
// and builds the cumulative table with a SEQUENTIAL partial_sum; the drawn indices depend on every rounding of those two
// chains, so they are kept sequential here -- one lane walks the million values -- and cost about what they cost the
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

// discrete_distribution::param_type::_M_initialize: sum = accumulate(p, 0.0); p /= sum; cp = partial_sum(p); cp.back() = 1.
// One wavefront: its 64 lanes move blocks of 4096 values between global memory and LDS (coalesced), lane 0 walks a block
// sequentially, sixteen LDS reads ahead of the dependent additions: 23 ms per step on a million transitions -- the chain of
// two million dependent fp64 additions of a lone wavefront (11 ns each; overlapping the LDS reads with the additions changes
// nothing; a first version passed the values lane to lane through v_readlane: 29 ms).  The CPU does the same chain in ~4 ms.
constexpr int PER_BLK = 4096;
__global__ __launch_bounds__(64) void per_scan_kernel(const float* __restrict__ prob, double* __restrict__ cp, long long n) {
  __shared__ double sh[PER_BLK];
  __shared__ double sSum;
  if (n < 2) return;                                   // (the distribution then always returns 0: nothing to build)
  const int lane = threadIdx.x;
  auto sync = [] { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };      // one wavefront: LDS traffic settled
  double s = 0.0;
  for (long long base = 0; base < n; base += PER_BLK) {
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) sh[j] = base + j < n ? (double)prob[base + j] : 0.0;    // (+ 0.0 behind the end: exact)
    sync();
    if (lane == 0) {
      for (int j = 0; j < PER_BLK; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sh[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
    }
    sync();
  }
  if (lane == 0) sSum = s;
  sync();
  s = sSum;
  double acc = 0.0;
  for (long long base = 0; base < n; base += PER_BLK) {
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) sh[j] = base + j < n ? (double)prob[base + j] / s : 0.0;
    sync();
    if (lane == 0) {
      for (int j = 0; j < PER_BLK; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sh[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc += v[u]; v[u] = acc; }
#pragma unroll
        for (int u = 0; u < 16; ++u) sh[j + u] = v[u];
      }
    }
    sync();
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) { const long long e = base + j; if (e < n) cp[e] = e == n - 1 ? 1.0 : sh[j]; }
    sync();
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
