// smarties_amd/csrc/sample.hip -- step tail: the minibatch sampler of step k+1 (workgroup 0) and
// the per-step bookkeeping of step k (workgroup 1) in ONE launch; the two are independent and
// run concurrently on two CUs.
//
//   sampler     Sample_uniform::sample + Sampling::IDtoSeqStep (Sampling.cpp:26-47,82-96) over
//               std::mt19937 generators[0] with libstdc++'s uniform_int_distribution (Lemire), the
//               gather of MemoryBuffer::sampleMinibatch (MemoryBuffer.cpp:413-429) and the
//               per-Adam-step generator draw (Optimizer.cpp:139).
//   bookkeeping Episode::updateCumulative_atomic / updateValues_atomic (Episode.h:112-145) in
//               minibatch order, MemoryProcessing::updateTrainingStatistics scalars (:187-259),
//               updateCounters (:46-92), Adam beta_t bookkeeping (Optimizer.cpp:155-160), step
//               counter (Learner.cpp:130-133)
//
// Both are dependency chains of small phases, so they are organised to minimise workgroup
// barriers and exposed memory latency: ballot-based scans, a bitonic sort whose strides < 64 run
// inside a wavefront on registers, an interpolation guess into a one-record-per-position episode
// table (one 64-byte fetch resolves flat index -> episode, step, slot, truncation for equal-length
// episodes; bounded binary search otherwise), and per-sample slots kept in LDS for the gather.
#include "dev_common.h"

namespace hl {

#define SMAXB 1024
#define SMAXK (SMAXB / 256)

// ---------------------------------------------------------------------------------------------
// bookkeeping ("post") part
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void aggValues(float* ag, float oldV, float oldADV, float V, float Q) {
  const float oldQ = oldADV + oldV;
  ag[AGG_SUMQ2] += Q * Q - oldQ * oldQ;
  ag[AGG_SUMQ] += Q - oldQ;
  ag[AGG_MAXQ] = fmaxf(ag[AGG_MAXQ], Q);
  ag[AGG_MINQ] = fminf(ag[AGG_MINQ], Q);
}

__device__ void postPart(const PostArgs& a, long long* sFarDelta, unsigned* sMaxAbs) {
  DevScalars* sc = a.sc;
  const int tid = threadIdx.x, B = a.B;
  if (tid == 0) { *sFarDelta = 0; *sMaxAbs = 0u; }
  __syncthreads();
  if (a.mode & POST_AGG) {
    const float C = (float)sc->Cmax, invC = (float)sc->Cinv;
    for (int b = tid; b < B; b += 256) {
      const int e = a.bt.pEid[b];
      const int ePrev = b > 0 ? a.bt.pEid[b - 1] : -1;
      if (ePrev == e) continue;                           // not the leader of this episode's run
      float* ag = a.rp.epAgg + (size_t)e * AGG_N;
      const float Nf = (float)a.rp.epN[e];
      const float invN = 1 / Nf;
      float g[AGG_N];
#pragma unroll
      for (int q = 0; q < AGG_N; ++q) g[q] = ag[q];
      const long long before = farSteps(Nf, g[AGG_FRACFAR]);
      for (int j = b; j < B && (j == b || a.bt.pEid[j] == e); ++j) {
        // every load of sample j first, then the (order-sensitive) float updates
        const int nxt = a.bt.pNextOf[j];
        const float E = a.bt.newDQ[j], D = a.bt.newDKL[j], W = a.bt.newW[j], Vf = a.bt.newV[j];
        const float oW = a.bt.oldW[j], oE = a.bt.oldDQ[j], oD = a.bt.oldDKL[j], oV = a.bt.oldV[j], oA = a.bt.oldADV[j];
        if (nxt >= 0) {                                    // setValues(t+1, Vnext) comes first
          const float Vn = a.bt.nextV[j];
          aggValues(g, a.bt.oldNextV[j], a.bt.oldNextADV[j], Vn, Vn);
        }
        const float wasFar = (oW > C || oW < invC) ? 1.f : 0.f;
        const float isFar = (W > C || W < invC) ? 1.f : 0.f;
        g[AGG_AVGKL] += invN * (D - oD);
        g[AGG_FRACFAR] += invN * (isFar - wasFar);
        g[AGG_AVGSQERR] += invN * (E * E - oE * oE);
        g[AGG_MAXABSERR] = fmaxf(g[AGG_MAXABSERR], fabsf(E));
        aggValues(g, oV, oA, Vf, Vf);
      }
#pragma unroll
      for (int q = 0; q < AGG_N; ++q) ag[q] = g[q];
      const long long after = farSteps(Nf, g[AGG_FRACFAR]);
      if (after != before) atomicAdd((unsigned long long*)sFarDelta, (unsigned long long)(after - before));
      atomicMax(sMaxAbs, __float_as_uint(fmaxf(g[AGG_MAXABSERR], 0.f)));
    }
    __syncthreads();
    if (tid == 0) {
      sc->nFarTotal += *sFarDelta;
      sc->maxAbsErrAll = fmaxf(sc->maxAbsErrAll, __uint_as_float(*sMaxAbs));
      // updateTrainingStatistics: ReF-ER clip annealing for the NEXT sampling (:193-196)
      const long long k = sc->nGradSteps + 1;
      sc->Cmax = 1 + a.clipImpWeight / (1 + (double)k * a.epsAnneal);
      sc->Cinv = 1 / sc->Cmax;
      if (sc->Cmax <= 1) sc->nFarTotal = 0;
      sc->nFarStat = sc->nFarTotal; sc->cnt[2] = sc->nFarStat; sc->cnt[3] = sc->nTransitions;
    }
    __syncthreads();
  }
  if ((a.mode & (POST_BETA | POST_INIT)) && tid == 0) {
    // updateCounters (:46-92); with several replicas cnt[] holds the all-reduced counters
    const long long nFar = a.nRanks > 1 ? sc->cnt[2] : sc->nFarStat;
    const long long nStored = a.nRanks > 1 ? sc->cnt[3] : sc->nTransitions;
    const double fracOffPol = (double)nFar / (double)(nStored > 1 ? nStored : 1);
    const double nDataSize = fmax(a.maxObsGlobal, (double)nStored);
    const double learnRefer = 0.1 * a.batchGlobal / nDataSize;
    const double b0 = sc->beta, al0 = sc->alpha;
    const bool dec = fracOffPol > a.penalTol;
    sc->beta = dec ? (1 - fmin(learnRefer, b0)) * b0 : (1 - fmin(learnRefer, b0)) * b0 + fmin(learnRefer, 1 - b0);
    const bool decA = fabs(a.penalTol - fracOffPol) < 1e-3;
    sc->alpha = decA ? (1 - fmin(learnRefer, al0)) * al0 : (1 - fmin(learnRefer, al0)) * al0 + fmin(learnRefer, 1 - al0);
    if (a.mode & POST_BETA) {
      // stats.maxAbsError EMA (:239-240) uses the replica-local data size
      const double lrLoc = 0.1 * a.batchGlobal / fmax(a.maxObsGlobal, (double)sc->nTransitions);
      sc->maxAbsErrEMA += lrLoc * ((double)sc->maxAbsErrAll - sc->maxAbsErrEMA);
      sc->adam_bt1 *= 0.9; if (sc->adam_bt1 < (double)FLT_EPSILON) sc->adam_bt1 = 0;
      sc->adam_bt2 *= 0.999; if (sc->adam_bt2 < (double)FLT_EPSILON) sc->adam_bt2 = 0;
      sc->nStep += 1;
      sc->nGradSteps += 1;
      sc->postPending = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// mt19937 (state in LDS)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mtTemper(unsigned z) {
  z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
  return z;
}
__device__ __forceinline__ unsigned mtF(unsigned a, unsigned b) {
  const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ void mtTwist(unsigned* x, unsigned* xo) {
  const int tid = threadIdx.x;
  for (int k = tid; k < 624; k += 256) xo[k] = x[k];
  __syncthreads();
  if (tid < 227) x[tid] = xo[tid + 397] ^ mtF(xo[tid], xo[tid + 1]);
  __syncthreads();
  if (tid < 227) x[227 + tid] = x[tid] ^ mtF(xo[227 + tid], xo[228 + tid]);
  __syncthreads();
  if (tid < 169) x[454 + tid] = x[227 + tid] ^ mtF(xo[454 + tid], xo[455 + tid]);
  __syncthreads();
  if (tid == 0) x[623] = x[396] ^ mtF(xo[623], x[0]);
  __syncthreads();
}
// append n raw tempered words to raw[0..n); *pPos lives in LDS; all threads call
__device__ void mtDraw(unsigned* x, unsigned* xo, int* pPos, unsigned* raw, int n) {
  int done = 0;
  while (done < n) {
    int pos = *pPos;
    __syncthreads();
    if (pos >= 624) { mtTwist(x, xo); pos = 0; }
    const int take = min(n - done, 624 - pos);
    for (int i = threadIdx.x; i < take; i += 256) raw[done + i] = mtTemper(x[pos + i]);
    if (threadIdx.x == 0) *pPos = pos + take;
    __syncthreads();
    done += take;
  }
}

// exclusive scan of one flag per element, element index e = r*256 + tid (r < K); returns the total.
// Two barriers per row of 256 elements (wave ballot + 4 wave totals through LDS).
__device__ int scanRows(int K, const bool* flag, int* excl, int* sWave /*[4]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int r = 0; r < K; ++r) {
    const unsigned long long m = __ballot(flag[r]);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sWave[wave] = __popcll(m);
    __syncthreads();
    const int w0 = sWave[0], w1 = sWave[1], w2 = sWave[2], w3 = sWave[3];
    excl[r] = base + within + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    base += w0 + w1 + w2 + w3;
    __syncthreads();
  }
  return base;
}

// bitonic sort of vals[0..Bp) (u32, Bp = 256*K a power of two); strides < 64 inside a wavefront
__device__ __forceinline__ unsigned cmpx(unsigned key, unsigned other, bool lower, bool up) {
  const unsigned mn = min(key, other), mx = max(key, other);
  return (lower == up) ? mn : mx;
}
__device__ void waveLocalRounds(unsigned* vals, int Bp, int k, int jStart) {
  // every wave owns the 64-element blocks wave, wave+4, ...
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int blk = wave; blk * 64 < Bp; blk += 4) {
    const int i = blk * 64 + lane;
    unsigned key = vals[i];
    if (k <= 64) {
      for (int kk = 2; kk <= k; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1)
          key = cmpx(key, (unsigned)__shfl_xor((int)key, j, 64), (i & j) == 0, (i & kk) == 0);
    } else {
      for (int j = jStart; j > 0; j >>= 1)
        key = cmpx(key, (unsigned)__shfl_xor((int)key, j, 64), (i & j) == 0, (i & k) == 0);
    }
    vals[i] = key;
  }
}
__device__ void bitonicSort(unsigned* vals, int Bp) {
  __syncthreads();
  waveLocalRounds(vals, Bp, 64, 32);              // all stages k = 2..64
  for (int k = 128; k <= Bp; k <<= 1) {
    for (int j = k >> 1; j >= 64; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < Bp / 2; t += 256) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
        const unsigned a = vals[i], b = vals[i | j];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { vals[i] = b; vals[i | j] = a; }
      }
    }
    __syncthreads();
    waveLocalRounds(vals, Bp, k, 32);
  }
  __syncthreads();
}

__device__ void samplePart(const SampleArgs& a, unsigned* x, unsigned* xo, unsigned* raw, unsigned* vals,
                           long long* sSlot, int* sNextRow, int* sWave, int* sPos) {
  const int tid = threadIdx.x;
  DevScalars* sc = a.sc;
  const int B = a.B;
  int Bp = 256; while (Bp < B) Bp <<= 1;
  const int K = Bp / 256;
  for (int k = tid; k < 624; k += 256) x[k] = sc->rng[k];
  if (tid == 0) *sPos = (int)sc->rngPos;
  const unsigned long long nData = (unsigned long long)sc->nTransitions;
  const int nEp = (int)sc->nEpisodes;
  __syncthreads();

  if (a.flatGiven) {
    for (int i = tid; i < B; i += 256) vals[i] = (unsigned)a.flatGiven[i];
    __syncthreads();
  } else {
    const unsigned range = (unsigned)nData;
    const unsigned threshold = (0u - range) % range;
    int have = 0;                       // vals[0..have) = sorted unique prefix
    while (have < B) {
      // ---- draw B-have accepted values (Lemire rejection, words consumed in order) ----
      int filled = have;
      while (filled < B) {
        const int need = B - filled;
        mtDraw(x, xo, sPos, raw, need);
        bool fl[SMAXK]; int ex[SMAXK]; unsigned v[SMAXK];
        for (int r = 0; r < K; ++r) {
          const int i = r * 256 + tid;
          fl[r] = false; v[r] = 0;
          if (i < need) {
            const unsigned long long prod = (unsigned long long)raw[i] * (unsigned long long)range;
            fl[r] = (unsigned)prod >= threshold; v[r] = (unsigned)(prod >> 32);
          }
        }
        const int acc = scanRows(K, fl, ex, sWave);
        for (int r = 0; r < K; ++r) if (fl[r]) vals[filled + ex[r]] = v[r];
        filled += acc;
      }
      for (int i = B + tid; i < Bp; i += 256) vals[i] = 0xFFFFFFFFu;
      bitonicSort(vals, Bp);
      // ---- std::unique ----
      bool fl[SMAXK]; int ex[SMAXK]; unsigned v[SMAXK];
      for (int r = 0; r < K; ++r) {
        const int i = r * 256 + tid;
        v[r] = i < B ? vals[i] : 0u;
        fl[r] = i < B && (i == 0 || v[r] != vals[i - 1]);
      }
      const int nu = scanRows(K, fl, ex, sWave);     // (barriers inside: all reads of vals are done)
      for (int r = 0; r < K; ++r) if (fl[r]) vals[ex[r]] = v[r];
      __syncthreads();
      have = nu;
    }
  }
  // ---- the generator draws of AdamOptimizer::apply_update (one per reference thread) ----
  if (a.adamDraws > 0) mtDraw(x, xo, sPos, raw, a.adamDraws);
  // generator state back to HBM early: independent of everything below
  for (int k = tid; k < 624; k += 256) sc->rng[k] = x[k];

  // ---- IDtoSeqStep: interpolation guess into the per-position table, then binary search ----
  bool hasNext[SMAXK]; int nextIdx[SMAXK];
  for (int r = 0; r < K; ++r) {
    const int b = r * 256 + tid;
    hasNext[r] = false;
    if (b < B) {
      const long long f = (long long)vals[b];
      int k0 = (int)(((double)f * (double)nEp) / (double)nData);
      k0 = min(max(k0, 0), nEp - 1);
      PosRec rec = a.rp.posRec[k0];
      const long long p1 = a.rp.posRec[k0 + 1].prefix;
      int lo = k0;
      if (f < rec.prefix || f >= p1) {
        int l = f < rec.prefix ? 0 : k0 + 1, hgh = f < rec.prefix ? k0 : nEp;   // largest k in [l,hgh): prefix[k] <= f
        while (hgh - l > 1) { const int mid = (l + hgh) >> 1; if (a.rp.posRec[mid].prefix <= f) l = mid; else hgh = mid; }
        lo = l; rec = a.rp.posRec[lo];
      }
      const int t = (int)(f - rec.prefix);
      const int e = rec.eidTerm & 0x7fffffff;
      const bool term = rec.eidTerm < 0;
      a.bt.flat[b] = f; a.bt.pos[b] = lo; a.bt.eid[b] = e; a.bt.t[b] = t; a.bt.tag[b] = rec.tag;
      a.bt.slot[b] = rec.off + t; sSlot[b] = rec.off + t;
      hasNext[r] = (t + 2 == rec.N && !term);      // Episode::isTruncated(t+1) (Episode.h:158-161)
    }
  }
  const int nNext = scanRows(K, hasNext, nextIdx, sWave);
  for (int r = 0; r < K; ++r) {
    const int b = r * 256 + tid;
    if (b < B) {
      if (hasNext[r]) { a.bt.nextOf[b] = B + nextIdx[r]; a.bt.nextSrc[nextIdx[r]] = b; sNextRow[b] = B + nextIdx[r]; }
      else { a.bt.nextOf[b] = -1; sNextRow[b] = -1; }
    }
  }
  if (tid == 0) { sc->rngPos = (unsigned)*sPos; sc->nNext = nNext; sc->nRows = B + nNext; }
  __syncthreads();
  // ---- gather: Episode::standardizedState (Episode.h:172-183) for s_t and truncated s_{t+1} ----
  const int dS = a.dS, total = B * dS;
  for (int e0 = tid; e0 < total; e0 += 256 * 8) {
    float sv[8], mv[8], cv[8]; int bb[8], ii[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      bb[u] = -1; sv[u] = 0.f; mv[u] = 0.f; cv[u] = 0.f; ii[u] = 0;
      if (e < total) {
        const int b = e / dS; bb[u] = b; ii[u] = e - b * dS;
        sv[u] = a.rp.S[(size_t)sSlot[b] * dS + ii[u]]; mv[u] = a.rp.stMean[ii[u]]; cv[u] = a.rp.stScale[ii[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) if (bb[u] >= 0) a.X0[(size_t)bb[u] * a.ldX0 + ii[u]] = (sv[u] - mv[u]) * cv[u];
  }
  if (nNext > 0) {
    for (int e = tid; e < total; e += 256) {
      const int b = e / dS, i = e - b * dS;
      const int nr = sNextRow[b];
      if (nr >= 0)
        a.X0[(size_t)nr * a.ldX0 + i] = (a.rp.S[(size_t)(sSlot[b] + 1) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
    }
  }
}

struct TailArgs { PostArgs post; SampleArgs samp; int doPost, doSample; };

// role of a workgroup: the sampler when both parts are requested and blockIdx == 0 (or alone),
// the bookkeeping pass otherwise.  doPost == 2: only if a trained step is pending (replayed graph:
// the same node serves the first step of an hl_step() call and all the following ones).
__global__ __launch_bounds__(256) void step_tail_kernel(TailArgs ta) {
  __shared__ unsigned x[624], xo[624];
  __shared__ unsigned raw[SMAXB];
  __shared__ unsigned vals[SMAXB];
  __shared__ long long sSlot[SMAXB];
  __shared__ int sNextRow[SMAXB];
  __shared__ int sWave[4];
  __shared__ int sPos;
  __shared__ long long sFarDelta;
  __shared__ unsigned sMaxAbs;
  const bool sampler = ta.doSample && (!ta.doPost || blockIdx.x == 0);
  if (sampler) {
    samplePart(ta.samp, x, xo, raw, vals, sSlot, sNextRow, sWave, &sPos);
  } else if (ta.doPost == 1 || (ta.doPost == 2 && ta.post.sc->postPending)) {
    postPart(ta.post, &sFarDelta, &sMaxAbs);
  }
}

hipError_t launch_step_tail(const PostArgs* post, const SampleArgs* samp, int postIfPending, hipStream_t s) {
  TailArgs ta{};
  ta.doPost = post ? (postIfPending ? 2 : 1) : 0; ta.doSample = samp != nullptr;
  if (post) ta.post = *post;
  if (samp) ta.samp = *samp;
  const int blocks = (post && samp) ? 2 : 1;
  hipLaunchKernelGGL(step_tail_kernel, dim3(blocks), dim3(256), 0, s, ta);
  return hipGetLastError();
}
hipError_t launch_sample(const SampleArgs& a, hipStream_t s) { return launch_step_tail(nullptr, &a, 0, s); }
hipError_t launch_post(const PostArgs& a, hipStream_t s) { return launch_step_tail(&a, nullptr, 0, s); }

}  // namespace hl
