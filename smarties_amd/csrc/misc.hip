// smarties_amd/csrc/misc.hip -- Adam (multi-replica path), per-step bookkeeping, periodic
// whole-buffer sweeps (Retrace, episode aggregates, reward/state moments) and statistics.
#include "dev_common.h"

namespace hl {

// ---------------------------------------------------------------------------
// adam_kernel: Adam::step + AdamOptimizer::apply_update (Network/Optimizer.cpp:61-108,122-160)
// with SMARTIES_NESTEROV_ADAM, SMARTIES_SAFE_ADAM, SMARTIES_ADAMW (Settings/Bund.h).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  AdamCoef c; c.eta = a.sc->etaEff[a.parity]; c.lambda = a.lambda; c.fac = a.fac;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    float w = a.W[i], m1 = a.M1[i], m2 = a.M2[i];
    adamStep(c, a.G[i], w, m1, m2);
    a.W[i] = w; a.M1[i] = m1; a.M2[i] = m2;
  }
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t s) {
  const int blocks = (int)((a.n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// episode_sweep_kernel: one WAVEFRONT per episode (grid-stride over the episodes).
//   recompute=1: Episode::updateCumulative (Episode.cpp:213-242)
//   then computeRetrace backward scan (MemoryProcessing.cpp:23-44,391-400).
// Used on insert, by initializeLearner and every 1000th step (whole buffer).
//
// Both passes are sequential recurrences in fp32 whose order the reference fixes, so they cannot be
// tree-reduced.  The 64 lanes load 64 consecutive steps with coalesced reads; the recurrence then
// runs over the chunk with v_readlane broadcasts (every lane carries the same accumulator), and the
// lane whose step it is keeps the value it has to store.  A thread per episode walked the arrays
// with a stride of one episode: every 4-byte read pulled its own cache line 16 times (440 us for
// 1M transitions); this is one coalesced pass (~10x less).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float rlF(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
__device__ __forceinline__ double rlD(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, l), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void episode_sweep_kernel(EpisodeSweepArgs a) {
  __shared__ float sMax[4];
  __shared__ double sErr[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nWaves = gridDim.x * 4;
  const DevScalars* sc = a.sc;
  float myMax = 0.f; double myErr = 0;
  // computeRetraceExplBonus (:402-408): coefficient 1 - gamma, baseline = ReplayStats::maxAbsError as it stands when the
  // sweep starts (createReturnEstimator captures it before this step's update, :431)
  const int kind = a.retKind;
  const float explCoef = 1 - a.gamma, explBase = (float)sc->maxAbsErrEMA;
  const float C = (float)sc->Cmax, invC = (float)sc->Cinv;
  const float gamma = a.gamma, lambda = a.lambda;
  const float rM = sc->rewMean, rS = sc->rewScale;
  for (int idx = blockIdx.x * 4 + wave; idx < a.count; idx += nWaves) {
    const int e = a.eids ? a.eids[idx] : a.rp.posEid[idx];
    const long long off = a.rp.epOff[e];
    const int N = a.rp.epN[e];
    const bool term = a.rp.epTerm[e] != 0;
    if (a.recompute) {
      const int nd = N - 1;
      const float invN = 1 / (float)nd;
      long long nFarPol = 0;
      float sumE2 = 0, maxAE = -1e9f, maxQ = -1e9f, sumQ2 = 0, minQ = 1e9f, sumQ1 = 0, sumKL = 0;
      double totR = 0;
      for (int t0 = 0; t0 < N; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < N;
        const float w = ok ? a.rp.IMPW[off + t] : 1.f, dq = ok ? a.rp.DQ[off + t] : 0.f;
        const float ad = ok ? a.rp.ADV[off + t] : 0.f, vv = ok ? a.rp.V[off + t] : 0.f;
        const double rr = ok ? a.rp.R[off + t] : 0.0; const float kk = ok ? a.rp.DKL[off + t] : 0.f;
        const int cnt = min(64, N - t0);
#pragma unroll 8
        for (int u = 0; u < cnt; ++u) {
          const float wu = rlF(w, u), dqu = rlF(dq, u), Q = rlF(ad, u) + rlF(vv, u);
          if (t0 + u < nd) {
            if (wu > C || wu < invC) ++nFarPol;
            sumE2 += dqu * dqu; maxAE = fmaxf(maxAE, fabsf(dqu));
            maxQ = fmaxf(maxQ, Q); minQ = fminf(minQ, Q); sumQ2 += Q * Q; sumQ1 += Q;
          }
          totR += rlD(rr, u); sumKL += rlF(kk, u);
        }
      }
      if (lane == 0) {
        float* ag = a.rp.epAgg + (size_t)e * AGG_N;
        ag[AGG_FRACFAR] = invN * (float)nFarPol; ag[AGG_AVGSQERR] = invN * sumE2; ag[AGG_MAXABSERR] = maxAE;
        ag[AGG_SUMQ2] = sumQ2; ag[AGG_SUMQ] = sumQ1; ag[AGG_MAXQ] = maxQ; ag[AGG_MINQ] = minQ;
        ag[AGG_TOTR] = (float)totR; ag[AGG_AVGKL] = invN * sumKL;
      }
      myMax = fmaxf(myMax, fmaxf(maxAE, 0.f));
    }
    if (a.skipRetrace || kind == HL_RET_NONE) continue;
    float Q = term ? a.rp.RET[off + N - 1] : a.rp.V[off + N - 1];
    if (!term && lane == 0) a.rp.RET[off + N - 1] = Q;
    float epErr = 0.f;                               // updateReturnEstimator's sumErr2 (Fval, each term added in double)
    for (int t1 = N - 2; t1 >= 0; t1 -= 64) {      // chunk covers t = t1, t1-1, ..., t1-63
      const int t = t1 - lane;
      const bool ok = t >= 0;
      const double rr = ok ? a.rp.R[off + t + 1] : 0.0;
      const float vv = ok ? a.rp.V[off + t + 1] : 0.f, ad = ok ? a.rp.ADV[off + t + 1] : 0.f, iw = ok ? a.rp.IMPW[off + t + 1] : 0.f;
      const float old = (ok && a.recompute) ? a.rp.RET[off + t] : 0.f;
      const float Rl = (float)((rr - (double)rM) * (double)rS);      // per lane, same expression as the reference
      const float wl = iw < 1.f ? iw : 1.f;
      const int cnt = min(64, t1 + 1);
      float mine = 0.f;
      if (kind == HL_RET_RETRACE) {
#pragma unroll 8
        for (int u = 0; u < cnt; ++u) {
          const float Vu = rlF(vv, u), Au = rlF(ad, u);
          Q = rlF(Rl, u) + gamma * (Vu + lambda * rlF(wl, u) * (Q - Au - Vu));
          if (lane == u) mine = Q;
        }
      } else if (kind == HL_RET_GAE) {
#pragma unroll 8
        for (int u = 0; u < cnt; ++u) {
          const float Vu = rlF(vv, u);
          Q = rlF(Rl, u) + gamma * (Vu + lambda * (Q - Vu));
          if (lane == u) mine = Q;
        }
      } else {
#pragma unroll 8
        for (int u = 0; u < cnt; ++u) {
          const float Vu = rlF(vv, u), Au = rlF(ad, u);
          const float dl = Q - Au - Vu;
          const float ret = rlF(Rl, u) + gamma * (Vu + lambda * rlF(wl, u) * dl);
          Q = explCoef * (fabsf(dl) - explBase) + ret;
          if (lane == u) mine = Q;
        }
      }
      if (ok) a.rp.RET[off + t] = mine;
      if (a.recompute) {
        const float df = old - mine;
        for (int u = 0; u < cnt; ++u) { const float du = rlF(df, u); epErr = (float)((double)epErr + (double)du * (double)du); }
      }
    }
    myErr += (double)epErr;
  }
  if (a.recompute) {
    if (lane == 0) { sMax[wave] = myMax; sErr[wave] = myErr; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a.redMaxAbs[blockIdx.x] = fmaxf(fmaxf(sMax[0], sMax[1]), fmaxf(sMax[2], sMax[3]));
      a.redErr[blockIdx.x] = (sErr[0] + sErr[1]) + (sErr[2] + sErr[3]);
    }
  }
}
int sweep_blocks(int count) { const int b = (count + 3) / 4; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }
hipError_t launch_episode_sweep(const EpisodeSweepArgs& a, int nBlocks, hipStream_t s) {
  if (nBlocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(episode_sweep_kernel, dim3(nBlocks), dim3(256), 0, s, a);
  return hipGetLastError();
}
// countRet >= 0: the sweep rewrote that many return estimates (MemoryProcessing.cpp:250-258); < 0: it did not touch them
__global__ __launch_bounds__(256) void sweep_finish_kernel(DevScalars* sc, DevReplay rp, const float* redMaxAbs, const double* redErr, int countRet, int n) {
  __shared__ unsigned long long sScan[4]; __shared__ float sMax[4]; __shared__ double sErr[4];
  const int tid = threadIdx.x;
  float m = 0.f; double er = 0;
  for (int i = tid; i < n; i += 256) { m = fmaxf(m, redMaxAbs[i]); er += redErr[i]; }
  for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); er += __shfl_xor(er, o, 64); }
  if ((tid & 63) == 0) { sMax[tid >> 6] = m; sErr[tid >> 6] = er; }
  __syncthreads();
  // the far-policy count as the reference's loop over the episodes accumulates it (dev_common.h; far_build_kernel ran before)
  const int nEp = (int)sc->nEpisodes, per = (nEp + 255) / 256, cnt = farSegment(nEp, per);
  const unsigned long long f = farFixedPoint([&](unsigned long long n0) { return farWalkMem<false>(rp.farP, rp.farN, cnt, n0); }, rp.farStart, sScan);
  if (tid != 0) return;
  m = fmaxf(fmaxf(sMax[0], sMax[1]), fmaxf(sMax[2], sMax[3])); er = (sErr[0] + sErr[1]) + (sErr[2] + sErr[3]);
  if (countRet >= 0) { sc->cntRetUpd = (sc->cntRetUpd < 0 ? 0 : sc->cntRetUpd) + countRet; sc->sumRetErr += er; }
  sc->nFarTotal = sc->Cmax <= 1 ? 0 : (long long)f;
  sc->maxAbsErrAll = m; sc->maxAbsErrStep = m;
  sc->nFarStat = sc->nFarTotal; sc->cnt[2] = sc->nFarStat; sc->cnt[3] = sc->nTransitions;
  sc->cnt[0] = sc->seenLocal[0]; sc->cnt[1] = sc->seenLocal[1];
}
// the terms of the far-policy count in the walk's layout (dev_common.h), for the table and the fractions as they stand
__global__ __launch_bounds__(256) void far_build_kernel(DevReplay rp, int nEp) {
  const int per = (nEp + 255) / 256;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= per * 256) {      // rows behind the table up to what the register walk reads (FAR_REGS per thread): zero terms
    if (pos < FAR_REGS * 256) { rp.farP[pos] = 0.f; rp.farN[pos] = 0.f; }
    return;
  }
  const int t = pos / per, i = pos - t * per;
  float f = 0.f, l = 0.f;
  if (pos < nEp) { const int e = rp.posEid[pos]; l = (float)rp.epN[e]; f = rp.epAgg[(size_t)e * AGG_N + AGG_FRACFAR]; }
  rp.farP[(size_t)i * 256 + t] = f; rp.farN[(size_t)i * 256 + t] = l;
}
hipError_t launch_far_build(DevReplay rp, int nEpisodes, hipStream_t s) {
  const int nb = (nEpisodes + 255) / 256 + 1;
  hipLaunchKernelGGL(far_build_kernel, dim3(nb > FAR_REGS ? nb : FAR_REGS), dim3(256), 0, s, rp, nEpisodes);
  return hipGetLastError();
}
hipError_t launch_sweep_finish(DevScalars* sc, DevReplay rp, const float* redMaxAbs, const double* redErr, int countRet, int nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(sweep_finish_kernel, dim3(1), dim3(256), 0, s, sc, rp, redMaxAbs, redErr, countRet, nBlocks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// moments: MemoryProcessing::updateRewardsStats (MemoryProcessing.cpp:94-185).
// One wavefront per episode (lanes stride over its transitions), fp64 accumulation
// (the reference uses long double on the host), deterministic two-stage reduction.
// ---------------------------------------------------------------------------
// Thread layout: column c = tid % (dS+1) (c < dS: state component, c == dS: reward), row lane
// r = tid / (dS+1); every thread owns one column, so sums are order-deterministic.
__global__ __launch_bounds__(256) void moments_partial_kernel(MomentsArgs a) {
  __shared__ double s1[256], s2[256];
  // wide states (dS + 1 > 256, e.g. Humanoid's 257): blockIdx.y selects a chunk of 256 columns
  const int dS = a.dS, CW = dS + 1, c0 = blockIdx.y * 256, CC = min(CW - c0, 256), RL = 256 / CC, tid = threadIdx.x;
  const int cl = tid % CC, c = c0 + cl, r = tid / CC;
  const bool active = r < RL;
  double sum = 0, sq = 0;
  const float rMean = a.sc->rewMean;
  const float sMean = (active && c < dS) ? a.rp.stMean[c] : 0.f;
  if (active) for (int p = blockIdx.x; p < a.nEpisodes; p += gridDim.x) {
    const int e = a.rp.posEid[p];
    const long long off = a.rp.epOff[e];
    const int nd = a.rp.epN[e] - 1;
    for (int j = r; j < nd; j += RL) {
      double d;
      if (c < dS) d = (double)(a.rp.S[(size_t)(off + j) * dS + c] - sMean);   // float - float
      else d = a.rp.R[off + j + 1] - (double)rMean;
      sum += d; sq += d * d;
    }
  }
  s1[tid] = sum; s2[tid] = sq;
  __syncthreads();
  if (tid < CC) {
    double t1 = 0, t2 = 0;
    for (int q = 0; q < RL; ++q) { t1 += s1[q * CC + tid]; t2 += s2[q * CC + tid]; }
    a.partial[(size_t)blockIdx.x * 2 * CW + c0 + tid] = t1;
    a.partial[(size_t)blockIdx.x * 2 * CW + CW + c0 + tid] = t2;
  }
}
// moments layout (MemoryProcessing.cpp:139-150): [sum s (dS) | sum s^2 (dS) | count | sum r | sum r^2]
// one wavefront per output: lanes stride over the per-block partials, then a butterfly sum
__global__ __launch_bounds__(256) void moments_final_kernel(MomentsArgs a) {
  const int dS = a.dS, CW = dS + 1;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i < 2 * CW) {
    double s = 0;
    for (int b = lane; b < a.nBlocks; b += 64) s += a.partial[(size_t)b * 2 * CW + i];
    s = waveSum(s);
    if (lane == 0) {
      const int c = i % CW; const bool second = i >= CW;
      if (c < dS) a.moments[(second ? dS : 0) + c] = s;
      else a.moments[2 * dS + (second ? 2 : 1)] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.moments[2 * dS] = (double)a.sc->nTransitions;
}
__global__ void moments_apply_kernel(MomentsArgs a) {
  DevScalars* sc = a.sc;
  const int dS = a.dS;
  const double learnR = a.learnrate / (1 + (double)sc->nGradSteps * a.epsAnneal);
  const double annealLearnR = fmin(1.0, a.rRateFac * learnR);
  const double Wt = a.bInit ? 1.0 : annealLearnR;
  if (!(Wt > 0)) return;
  const double count = a.moments[2 * dS];
  for (int i = threadIdx.x; i <= dS; i += blockDim.x) {
    const bool isRew = (i == dS);
    const double Evar = (isRew ? a.moments[2 * dS + 1] : a.moments[i]) / count;
    const double Evar2 = (isRew ? a.moments[2 * dS + 2] : a.moments[dS + i]) / count;
    float mean = isRew ? sc->rewMean : a.rp.stMean[i];
    float stdev = isRew ? sc->rewStd : a.rp.stStd[i];
    mean = (float)((double)mean + Wt * Evar);
    double variance = Evar2 - Evar * Evar * (2 * Wt - Wt * Wt);
    variance = fmax(variance, (double)FLT_EPSILON);
    stdev = (float)((double)stdev + Wt * (sqrt(variance) - (double)stdev));
    const float inv = 1 / stdev;
    if (isRew) { sc->rewMean = mean; sc->rewStd = stdev; sc->rewScale = inv; }
    else { a.rp.stMean[i] = mean; a.rp.stStd[i] = stdev; a.rp.stScale[i] = inv; }
  }
}
int moments_blocks(int nEpisodes) { int b = nEpisodes; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }
hipError_t launch_moments(const MomentsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(moments_partial_kernel, dim3(a.nBlocks, (a.dS + 1 + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(moments_final_kernel, dim3((2 * (a.dS + 1) + 3) / 4), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_moments_apply(const MomentsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(moments_apply_kernel, dim3(1), dim3(128), 0, s, a);
  return hipGetLastError();
}

__global__ void set_counts_kernel(DevScalars* sc, long long nT, long long nE, long long seenEps, long long seenSteps) {
  sc->nTransitions = nT; sc->nEpisodes = nE; sc->cnt[0] = seenEps; sc->cnt[1] = seenSteps;
  sc->seenLocal[0] = seenEps; sc->seenLocal[1] = seenSteps;
  sc->cnt[3] = nT;   // cnt[2] keeps the far-policy count of the last statistics pass
}
// max over the stored episodes of Episode::maxAbsError (one workgroup; after arrivals / removals)
__global__ __launch_bounds__(256) void episode_max_kernel(DevScalars* sc, DevReplay rp, int nEp) {
  __shared__ float sm[4];
  float m = 0.f;
  for (int p = threadIdx.x; p < nEp; p += 256) m = fmaxf(m, rp.epAgg[(size_t)rp.posEid[p] * AGG_N + AGG_MAXABSERR]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) sc->maxAbsErrAll = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
hipError_t launch_episode_max(DevScalars* sc, DevReplay rp, int nEp, hipStream_t s) {
  hipLaunchKernelGGL(episode_max_kernel, dim3(1), dim3(256), 0, s, sc, rp, nEp);
  return hipGetLastError();
}
hipError_t launch_set_counts(DevScalars* sc, long long nT, long long nE, long long seenEps, long long seenSteps, hipStream_t s) {
  hipLaunchKernelGGL(set_counts_kernel, dim3(1), dim3(1), 0, s, sc, nT, nE, seenEps, seenSteps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stats_kernel (one workgroup): the reduction over all episodes of
// MemoryProcessing::updateTrainingStatistics (:209-258), on demand (logging surface).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_kernel(DevScalars* sc, DevReplay rp, int nEp, double* out) {
  __shared__ double sd[5][256];
  __shared__ float sf[2][256];
  const int tid = threadIdx.x;
  double sumDKL = 0, sumE2 = 0, sumQ2 = 0, sumQ1 = 0, sumR = 0;
  float maxQ = -1e9f, minQ = 1e9f;
  for (int p = tid; p < nEp; p += 256) {
    const int e = rp.posEid[p];
    const float* ag = rp.epAgg + (size_t)e * AGG_N;
    const float Nf = (float)rp.epN[e];
    sumDKL += (double)(Nf * ag[AGG_AVGKL]); sumE2 += (double)(Nf * ag[AGG_AVGSQERR]);
    sumQ2 += (double)ag[AGG_SUMQ2]; sumQ1 += (double)ag[AGG_SUMQ]; sumR += (double)ag[AGG_TOTR];
    maxQ = fmaxf(maxQ, ag[AGG_MAXQ]); minQ = fminf(minQ, ag[AGG_MINQ]);
  }
  sd[0][tid] = sumDKL; sd[1][tid] = sumE2; sd[2][tid] = sumQ2; sd[3][tid] = sumQ1; sd[4][tid] = sumR;
  sf[0][tid] = maxQ; sf[1][tid] = minQ;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      for (int q = 0; q < 5; ++q) sd[q][tid] += sd[q][tid + s];
      sf[0][tid] = fmaxf(sf[0][tid], sf[0][tid + s]); sf[1][tid] = fminf(sf[1][tid], sf[1][tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double nData = fmax(1.0, (double)sc->nTransitions);
    out[0] = sd[0][0] / nData;                 // avgKLdivergence
    out[1] = sd[1][0] / nData;                 // avgSquaredErr
    out[2] = sc->maxAbsErrEMA;                 // maxAbsError
    out[3] = sd[4][0] / (double)nEp;           // avgReturn
    const double avgQ = sd[3][0] / nData;
    out[4] = avgQ;
    out[5] = sqrt(fmax(sd[2][0] / nData - avgQ * avgQ, 1e-16));   // stdevQ
    out[6] = (double)sf[1][0]; out[7] = (double)sf[0][0];          // minQ, maxQ
    out[8] = (double)sc->nFarStat;
  }
}
hipError_t launch_stats(DevScalars* sc, DevReplay rp, int nEpisodes, double* out, hipStream_t s) {
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(256), 0, s, sc, rp, nEpisodes, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// rollout inference (hl_forward): standardise raw states into the minibatch rows, run the forward
// GEMMs of the training path on them, then the Linear output layer + ParamLayer as doubles
// ---------------------------------------------------------------------------
// dIn = dS (1 + nAppendedObs): a row holds the raw state of step t followed by the appended past ones, each standardised
// with the per-component mean / scale (Episode::standardizedState, Episode.h:172-183)
__global__ __launch_bounds__(256) void act_standardize_kernel(DevScalars* sc, DevReplay rp, const float* S, int n, int dS, int dIn, float* X0, int ldX0) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { sc->nRows[0] = n; sc->nNext[0] = 0; }     // rows the forward GEMMs of buffer 0 will process
  if (i < (long long)n * dIn) { const int r = (int)(i / dIn), c = (int)(i - (long long)r * dIn), k = c % dS;
    X0[(size_t)r * ldX0 + c] = (S[i] - rp.stMean[k]) * rp.stScale[k]; }
}
// done != nullptr (one row, outputs in pinned host memory): the row is stamped once its outputs are visible to the host
__global__ __launch_bounds__(256) void act_output_kernel(const float* Y, int ldY, int H, const float* W, long long indWo, long long indBo,
                                                         long long indBp, int ldWo, int nDense, int dA, int n, double* O, unsigned* done, unsigned tag, int outFunc) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, nOut = nDense + dA;
  if (row >= n) return;
  for (int o = 0; o < nDense; ++o) {
    float p = 0.f;
    for (int k = lane; k < H; k += 64) p += Y[(size_t)row * ldY + k] * W[indWo + (long long)k * ldWo + o];
    p = waveSumF(p);
    if (lane == 0) O[(size_t)row * nOut + o] = (double)actEval(outFunc, p + W[indBo + o]);
  }
  if (lane < dA) O[(size_t)row * nOut + nDense + lane] = (double)W[indBp + lane];
  if (done) {
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(done + row, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The network for ONE raw state per workgroup: what an environment thread needs per agent step (RACER::selectAction,
// Learners/RACER.cpp:30-59 -> Approximator::forward(agent)).  The batched path above costs five launches and two staged copies
// (47 us per call); here the state comes straight from pinned host memory, every layer is a per-thread dot product over the
// previous layer's outputs in LDS (weights row-major: thread j reads W[k][j], coalesced), the outputs go back to pinned host
// memory and a per-row stamp tells the waiting host thread that they are there.
__global__ __launch_bounds__(256) void act_forward_kernel(ActArgs a) {
  __shared__ float sA[ACT_MAXW], sB[ACT_MAXW];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* W = a.W;
  for (int c = tid; c < a.dIn; c += 256) { const int k = c % a.dS; sA[c] = (a.in[(size_t)row * a.dIn + c] - a.stMean[k]) * a.stScale[k]; }
  __syncthreads();
  float* in = sA; float* out = sB;
  for (int l = 0; l < a.nL; ++l) {
    const ActLayer L = a.L[l];
    for (int j = tid; j < L.size; j += 256) {
      const float* w = W + L.indW + j;
      float acc0 = W[L.indB + j], acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      int k = 0;
      for (; k + 8 <= L.nIn; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w[(size_t)(k + u) * L.ldW];
        acc0 = fmaf(in[k], v[0], acc0); acc1 = fmaf(in[k + 1], v[1], acc1); acc2 = fmaf(in[k + 2], v[2], acc2); acc3 = fmaf(in[k + 3], v[3], acc3);
        acc0 = fmaf(in[k + 4], v[4], acc0); acc1 = fmaf(in[k + 5], v[5], acc1); acc2 = fmaf(in[k + 6], v[6], acc2); acc3 = fmaf(in[k + 7], v[7], acc3);
      }
      for (; k < L.nIn; ++k) acc0 = fmaf(in[k], w[(size_t)k * L.ldW], acc0);
      float y = actEval(L.func, (acc0 + acc1) + (acc2 + acc3));
      if (L.hasRes && j < L.resW) y += in[j] * W[L.indWr + j] + W[L.indBr + j];       // ParametricResidualLayer::forward (Layers.h:347-361)
      out[j] = y;
    }
    __syncthreads();
    float* t = in; in = out; out = t;
  }
  // output layer (Linear) + ParamLayer: one output per wavefront at a time, lanes over the hidden units
  const int H = a.L[a.nL - 1].size;
  for (int o = wave; o < a.nDense; o += 4) {
    float p = 0.f;
    for (int k = lane; k < H; k += 64) p = fmaf(in[k], W[a.indWo + (long long)k * a.ldWo + o], p);
    p = waveSumF(p);
    if (lane == 0) a.out[(size_t)row * a.nOut + o] = (double)actEval(a.outFunc, p + W[a.indBo + o]);
  }
  if (tid < a.nSig) a.out[(size_t)row * a.nOut + a.nDense + tid] = (double)W[a.indBp + tid];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(const_cast<unsigned*>(a.done) + row, a.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_act_forward(const ActArgs& a, int n, hipStream_t s) {
  hipLaunchKernelGGL(act_forward_kernel, dim3(n), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_act_standardize(DevScalars* sc, DevReplay rp, const float* S, int n, int dS, int dIn, float* X0, int ldX0, hipStream_t s) {
  hipLaunchKernelGGL(act_standardize_kernel, dim3((unsigned)(((long long)n * dIn + 255) / 256 + 1)), dim3(256), 0, s, sc, rp, S, n, dS, dIn, X0, ldX0);
  return hipGetLastError();
}
hipError_t launch_act_output(const float* Y, int ldY, int H, const float* W, long long indWo, long long indBo, long long indBp, int ldWo,
                             int nDense, int dA, int n, double* O, hipStream_t s, unsigned* done, unsigned tag, int outFunc) {
  hipLaunchKernelGGL(act_output_kernel, dim3((n + 3) / 4), dim3(256), 0, s, Y, ldY, H, W, indWo, indBo, indBp, ldWo, nDense, dA, n, O, done, tag, outFunc);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// impw_hist_kernel: MemoryProcessing::histogramImportanceWeights (MemoryProcessing.cpp:353-389): one wavefront per
// episode, lanes over its transitions, bins found by bisection of the (monotonic) bounds, block histogram in LDS
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void impw_hist_kernel(HistArgs a) {
  __shared__ unsigned sCnt[81];
  __shared__ float sB[82];
  for (int i = threadIdx.x; i < 82; i += 256) { sB[i] = a.bounds[i]; if (i < 81) sCnt[i] = 0u; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = blockIdx.x * 4 + wave; p < a.nEpisodes; p += gridDim.x * 4) {
    const int e = a.rp.posEid[p];
    const long long off = a.rp.epOff[e];
    const int nd = a.rp.epN[e] - 1;
    for (int j = lane; j < nd; j += 64) {
      const float rho = a.rp.IMPW[off + j];
      if (rho >= sB[0] && rho < sB[81]) {
        int lo = 0, hi = 81;                       // largest b with bounds[b] <= rho
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sB[mid] <= rho) lo = mid; else hi = mid; }
        atomicAdd(&sCnt[lo], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 81; i += 256) if (sCnt[i]) atomicAdd(&a.counts[i], (unsigned long long)sCnt[i]);
}
hipError_t launch_impw_hist(const HistArgs& a, hipStream_t s) {
  int nb = (a.nEpisodes + 3) / 4; if (nb > 512) nb = 512; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(impw_hist_kernel, dim3(nb), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ingest_kernel: MemoryBuffer::addEpisodeToTrainingSet / pushBackEpisode (MemoryBuffer.cpp:131-170, 479-520) for a batch of
// episodes: blockIdx.y = episode of the batch, blockIdx.x strides over its elements.  Reads the pinned host buffer over the
// bus exactly once; the derived per-step fields get their insertion values (Episode.h:66-82; pre-training error placeholder
// sqrt(max(eps, avgSquaredErr)), MemoryBuffer.cpp:486-487), the Retrace estimate follows from the episode sweep.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a) {
  const IngestDesc d = reinterpret_cast<const IngestDesc*>(a.stage)[blockIdx.y];
  const int N = d.N, dS = a.dS, dA = a.dA, pD = a.polDim;
  const unsigned char* base = a.stage + d.data;
  auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const float* S = reinterpret_cast<const float*>(base);
  const double* A = reinterpret_cast<const double*>(base + al16((size_t)N * dS * 4));
  const double* MU = A + (size_t)N * dA;                       // (multiples of 8 bytes keep the following blocks aligned)
  const double* R = MU + (size_t)N * pD;
  const float* V = reinterpret_cast<const float*>(R + N);
  const float* ADV = V + N;
  const double avgSq = a.nEpTable > 0 ? a.stats[1] : 0.0;
  const float maxError = (float)sqrt(fmax((double)FLT_EPSILON, avgSq));      // (Fval) std::sqrt(std::max(EPS, stats.avgSquaredErr))
  const long long off = d.off;
  const int tid = blockIdx.x * 256 + threadIdx.x, nT = gridDim.x * 256;
  if ((((size_t)N * dS) & 3) == 0 && ((off * dS) & 3) == 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(S); f32x4* d4 = reinterpret_cast<f32x4*>(a.rp.S + (size_t)off * dS);
    for (int i = tid; i < (N * dS) >> 2; i += nT) d4[i] = s4[i];
  } else for (int i = tid; i < N * dS; i += nT) a.rp.S[(size_t)off * dS + i] = S[i];
  for (int i = tid; i < N * dA; i += nT) a.rp.A[(size_t)off * dA + i] = A[i];
  for (int i = tid; i < N * pD; i += nT) a.rp.MU[(size_t)off * pD + i] = MU[i];
  for (int t = tid; t < N; t += nT) {
    a.rp.R[off + t] = R[t]; a.rp.V[off + t] = V[t]; a.rp.ADV[off + t] = ADV[t];
    a.rp.RET[off + t] = 0.f; a.rp.DQ[off + t] = maxError; a.rp.IMPW[off + t] = t == N - 1 ? 0.f : 1.f; a.rp.DKL[off + t] = 0.f;
  }
  if (tid == 0) {
    a.rp.epOff[d.eid] = off; a.rp.epN[d.eid] = N; a.rp.epTerm[d.eid] = d.term ? 1 : 0; a.rp.epTag[d.eid] = d.tag;
    float* ag = a.rp.epAgg + (size_t)d.eid * AGG_N;
    for (int q = 0; q < AGG_N; ++q) ag[q] = 0.f;
    ag[AGG_TOTR] = d.totR; ag[AGG_AVGSQERR] = maxError * maxError; ag[AGG_MAXABSERR] = maxError; ag[AGG_MAXQ] = -1e9f; ag[AGG_MINQ] = 1e9f;
  }
}
hipError_t launch_ingest(const IngestArgs& a, hipStream_t s) {
  if (a.nEp <= 0) return hipSuccess;
  hipLaunchKernelGGL(ingest_kernel, dim3(8, a.nEp), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void rng_restore_kernel(DevScalars* sc) {
  for (int k = threadIdx.x; k < 624; k += 256) sc->rng[k] = sc->rngBak[k];
  if (threadIdx.x == 0) sc->rngPos = sc->rngBakPos;
}
hipError_t launch_rng_restore(DevScalars* sc, hipStream_t s) { hipLaunchKernelGGL(rng_restore_kernel, dim3(1), dim3(256), 0, s, sc); return hipGetLastError(); }

// Completion stamp of a replayed call: the last node of an exact-size graph stores the count of such graphs run so far
// into pinned host memory; hl_sync polls that word instead of asking the runtime for a stream marker
// (tools/call_bench.hip: 7.5 us from the last workgroup to the host against 18.7 us through hipStreamSynchronize).
__global__ void notify_kernel(DevScalars* sc, unsigned* hostWord) {
  const unsigned s = ++sc->notifySeq;
  __hip_atomic_store(hostWord, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_notify(DevScalars* sc, unsigned* hostWord, hipStream_t s) { hipLaunchKernelGGL(notify_kernel, dim3(1), dim3(1), 0, s, sc, hostWord); return hipGetLastError(); }

// One word per 4 KB of every replay array: the address translations of the whole replay are resident before a timed
// stepping phase starts (hl_prepare_steps).  A minibatch touches ~2000 random rows of a few hundred MB; a freshly filled
// replay otherwise pays its page walks over the first few dozen steps (tools/first_call3.py).
__global__ __launch_bounds__(256) void touch_kernel(TouchArgs a) {
  // workgroups are dealt round-robin to the 8 XCDs: the 32 workgroups of one XCD together read every page, so that each XCD's
  // own translation caches have seen the whole replay (a minibatch row is gathered by whichever XCD its workgroup landed on)
  const int sub = blockIdx.x >> 3, nSub = gridDim.x >> 3;
  float acc = 0.f;
  for (int k = 0; k < a.n; ++k) {
    const char* base = (const char*)a.ptr[k];
    const long long st = a.stride[k];       // 4096: one word per page (translations); 64: every cache line (small tables: data resident)
    for (long long off = ((long long)sub * 256 + threadIdx.x) * st; off < a.bytes[k]; off += (long long)nSub * 256 * st)
      acc += *(const volatile float*)(base + off);
  }
  if (acc == 1.2345e-30f) *a.sink = acc;
}
hipError_t launch_touch(const TouchArgs& a, hipStream_t s) { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, s, a); return hipGetLastError(); }

__global__ void set_ret_counters_kernel(DevScalars* sc, long long cnt) { sc->cntRetUpd = cnt; sc->sumRetErr = 0; }
hipError_t launch_set_ret_counters(DevScalars* sc, long long cnt, hipStream_t s) { hipLaunchKernelGGL(set_ret_counters_kernel, dim3(1), dim3(1), 0, s, sc, cnt); return hipGetLastError(); }

__global__ void empty_kernel() {}
hipError_t launch_empty(hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); return hipGetLastError(); }

}  // namespace hl
