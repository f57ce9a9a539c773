// smarties_amd/csrc/tail_dev.h -- device code of the step tail: minibatch sampler and per-step
// bookkeeping.  Both are single-workgroup dependency chains.  They can run as their own kernel
// (sample.hip) or as ONE EXTRA WORKGROUP appended to the grid of an MLP kernel (gemm16.hip,
// head.hip): the sampler of step k+1 is cut into three phases that ride along fwd0(k), fwd1(k)
// and head(k), the bookkeeping of step k rides along dX(k) -- "horizontal fusion", which hides
// ~20 us of serial work per step behind kernels that leave 255 CUs idle anyway, without the
// cross-queue synchronisation a multi-stream graph costs on this runtime.
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
// Organised to minimise workgroup barriers and exposed memory latency: ballot-based scans, a
// bitonic sort whose strides < 64 run inside a wavefront on registers, an interpolation guess into
// a one-record-per-position episode table (one 64-byte fetch resolves flat index -> episode, step,
// slot, truncation for equal-length episodes; bounded binary search otherwise), and per-sample
// slots kept in LDS for the gather.
#pragma once
#include "dev_common.h"

namespace hl {

// development time stamps (100 MHz constant clock), enabled with -DHL_TAIL_STAMPS
#ifdef HL_TAIL_STAMPS
#define TSTAMP(sc, i) do { if (threadIdx.x == 0) (sc)->dbgT[i] = wall_clock64(); } while (0)
#define PSTAMP(sc, i) do { if (threadIdx.x == 0 && (a.mode & POST_DEFER)) (sc)->dbgT[i] = wall_clock64(); } while (0)      // (the steps inside a call, not its last)
#else
#define TSTAMP(sc, i) do { } while (0)
#define PSTAMP(sc, i) do { } while (0)
#endif

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

// the four replica counters as 16 floats: four 16-bit chunks each, so that an fp32 SUM all-reduce over up to 256
// replicas returns them exactly (every chunk sum stays below 2^24)
__device__ __forceinline__ void encodeCounters(float* msg, const long long c4[4]) {
  for (int c = 0; c < 4; ++c) for (int q = 0; q < 4; ++q) msg[4 * c + q] = (float)((c4[c] >> (16 * q)) & 0xFFFF);
}

// MemoryProcessing::updateCounters, the penalisation coefficients (:80-92); coherent: the stores are agent-scope atomics (read
// by workgroups of the same kernel on other XCDs)
__device__ __forceinline__ void refEerPenal(DevScalars* sc, double fracOffPol, double learnRefer, double penalTol, double beta0, double alpha0, bool coherent) {
  const bool dec = fracOffPol > penalTol;
  const double beta = dec ? (1 - fmin(learnRefer, beta0)) * beta0
                          : (1 - fmin(learnRefer, beta0)) * beta0 + fmin(learnRefer, 1 - beta0);
  const bool decA = fabs(penalTol - fracOffPol) < 1e-3;
  const double alpha = decA ? (1 - fmin(learnRefer, alpha0)) * alpha0
                            : (1 - fmin(learnRefer, alpha0)) * alpha0 + fmin(learnRefer, 1 - alpha0);
  if (coherent) {
    __hip_atomic_store(&sc->beta, beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc->alpha, alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else { sc->beta = beta; sc->alpha = alpha; }
}

// modeOv >= 0: the pass runs in that mode instead of a.mode (the folded weight-gradient launch closes its step with the record of its
// bookkeeping rider: no second PostArgs among the kernel arguments, no copy on the stack)
__device__ __forceinline__ void postPart(const PostArgs& a, long long* sFarDelta, unsigned* sMaxAbs, float* sFarP = nullptr, int farLdsFloats = 0, int modeOv = -1) {
  const int mode = modeOv >= 0 ? modeOv : a.mode;
  DevScalars* sc = a.sc;
  const int tid = threadIdx.x, B = a.B;
  // every scalar the pass needs, fetched once up front (uniform loads); thread 0 writes the
  // results back at the end without any further dependent global read
  const double Cmax0 = sc->Cmax, Cinv0 = sc->Cinv, beta0 = sc->beta, alpha0 = sc->alpha;
  const double ema0 = sc->maxAbsErrEMA, bt1 = sc->adam_bt1, bt2 = sc->adam_bt2;
  const long long nGrad0 = sc->nGradSteps, nFarTot0 = sc->nFarTotal, nFarStat0 = sc->nFarStat;
  const long long nTrans = sc->nTransitions, cnt0 = sc->cnt[0], cnt1 = sc->cnt[1], cnt2 = sc->cnt[2], cnt3 = sc->cnt[3], nStep0 = sc->nStep, nEpL = sc->nEpisodes;
  const float maxAll0 = (mode & POST_AGG) ? sc->maxAbsErrAll : sc->maxAbsErrStep;
  PSTAMP(sc, 16);
  if (tid == 0) { *sFarDelta = 0; *sMaxAbs = 0u; }
  // the terms of the far-policy count (dev_common.h): this thread's segment, fetched now, used after the aggregates are updated
  const int nEp = (int)nEpL, farPer = (nEp + 255) / 256;
  const bool defer = (mode & POST_DEFER) != 0;      // the count itself is taken by farBetaPhase: only the fractions are patched here
  // large batches (sample.hip: post_agg_chunks_kernel): aggChunk == 1 -- this workgroup updates the episode records of ITS 256 samples
  // (a run of samples of one episode belongs to the workgroup of its first sample) and leaves; aggChunk == 2 -- that has been done
  // by the launch in front, the maximum waits in DevScalars::maxAbsScratch
  const int chunkMode = a.aggChunk;
  const bool farOn = (mode & POST_AGG) != 0, farLds = farOn && !defer && !chunkMode && sFarP && farPer <= FAR_REGS && FAR_REGS * 256 <= farLdsFloats;
  float farT[FAR_REGS], farL[FAR_REGS];
  unsigned long long farG0[FAR_SUB] = {};
  if (farLds) {
#pragma unroll
    for (int c = 0; c < FAR_SUB; ++c) farG0[c] = a.rp.farStart[c * 256 + tid];
#pragma unroll
    for (int i = 0; i < FAR_REGS; ++i) { farT[i] = a.rp.farP[i * 256 + tid]; farL[i] = a.rp.farN[i * 256 + tid]; }      // (zeros behind the table: far_build_kernel)
  }
  if (!farLds) __syncthreads();       // (thread 0's initialisation above; with farLds the barrier in front of the leaders' work orders it)
  PSTAMP(sc, 17);
  long long nFarStat = nFarStat0; float maxAll = maxAll0;
  if (mode & POST_AGG) {
    const float C = (float)Cmax0, invC = (float)Cinv0;
    const int bBeg = chunkMode == 1 ? (int)blockIdx.x * 256 : 0, bEnd = chunkMode == 1 ? min(B, bBeg + 256) : (chunkMode == 2 ? 0 : B);
    for (int b0 = bBeg; b0 < bEnd; b0 += 256) {
      const int b = b0 + tid;
      const bool in = b < B;
      // round 1: everything indexed by the sample (coalesced), unconditionally
      const int e = in ? a.bt.pEid[b] : -1;
      const int ePrev = (in && b > 0) ? a.bt.pEid[b - 1] : -2;
      const int eNext = (in && b + 1 < B) ? a.bt.pEid[b + 1] : -3;
      const int nxt0 = in ? a.bt.pNextOf[b] : -1;
      float Q0 = 0, E0 = 0, D0 = 0, W0 = 0, V0 = 0, oW0 = 0, oE0 = 0, oD0 = 0, oV0 = 0, oA0 = 0, Vn0 = 0, oNV0 = 0, oNA0 = 0;
      if (in) {
        E0 = a.bt.newDQ[b]; D0 = a.bt.newDKL[b]; W0 = a.bt.newW[b]; V0 = a.bt.newV[b];
        oW0 = a.bt.oldW[b]; oE0 = a.bt.oldDQ[b]; oD0 = a.bt.oldDKL[b]; oV0 = a.bt.oldV[b]; oA0 = a.bt.oldADV[b];
        Vn0 = a.bt.nextV[b]; oNV0 = a.bt.oldNextV[b]; oNA0 = a.bt.oldNextADV[b];
        Q0 = a.hasAdv ? a.bt.newQ[b] : V0;          // VRACER: Q = V (no extra load on the hot path)
      }
      // the episode record: staged per sample by the fused kernel (same round as the loads above) ...
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
      if (in && a.aggStaged) {
        const f32x4* st = reinterpret_cast<const f32x4*>(a.bt.aggIn + (size_t)b * AGG_N);
        s0 = st[0]; s1 = st[1]; s2 = st[2];
      }
      const bool leader = in && ePrev != e;               // first sample of this episode's run
      float myMaxAbs = 0.f;
      if (b0 == 0 && farLds) {          // the terms go through LDS so that the leaders below can patch other threads' segments
#pragma unroll
        for (int i = 0; i < FAR_REGS; ++i) sFarP[i * 256 + tid] = farT[i];
        __syncthreads();
      }
      if (leader) {
      float* ag = a.rp.epAgg + (size_t)e * AGG_N;
      float g[AGG_N]; float Nf;
      if (a.aggStaged) {
        g[0] = s0[0]; g[1] = s0[1]; g[2] = s0[2]; g[3] = s0[3]; g[4] = s1[0]; g[5] = s1[1]; g[6] = s1[2]; g[7] = s1[3];
        g[8] = s2[0]; g[9] = 0.f; g[10] = 0.f; g[11] = 0.f; Nf = s2[1];
      } else {   // ... or a second, dependent round: gather it here
        Nf = (float)a.rp.epN[e];
#pragma unroll
        for (int q = 0; q < AGG_N; ++q) g[q] = ag[q];
      }
      const float invN = 1 / Nf;
      int j = b; bool more = true;
      float Qf = Q0, E = E0, D = D0, W = W0, Vf = V0, oW = oW0, oE = oE0, oD = oD0, oV = oV0, oA = oA0, Vn = Vn0, oNV = oNV0, oNA = oNA0;
      int nxt = nxt0;
      while (more) {
        if (nxt >= 0) aggValues(g, oNV, oNA, Vn, Vn);     // setValues(t+1, Vnext) comes first
        const float wasFar = (oW > C || oW < invC) ? 1.f : 0.f;
        const float isFar = (W > C || W < invC) ? 1.f : 0.f;
        g[AGG_AVGKL] += invN * (D - oD);
        g[AGG_FRACFAR] += invN * (isFar - wasFar);
        g[AGG_AVGSQERR] += invN * (E * E - oE * oE);
        g[AGG_MAXABSERR] = fmaxf(g[AGG_MAXABSERR], fabsf(E));
        aggValues(g, oV, oA, Vf, Qf);
        // further samples of the same episode (rare): fetched on demand, in minibatch order
        const int en = (j == b) ? eNext : ((j + 1 < B) ? a.bt.pEid[j + 1] : -3);
        more = (en == e);
        if (more) {
          ++j;
          nxt = a.bt.pNextOf[j];
          E = a.bt.newDQ[j]; D = a.bt.newDKL[j]; W = a.bt.newW[j]; Vf = a.bt.newV[j];
          oW = a.bt.oldW[j]; oE = a.bt.oldDQ[j]; oD = a.bt.oldDKL[j]; oV = a.bt.oldV[j]; oA = a.bt.oldADV[j];
          Vn = a.bt.nextV[j]; oNV = a.bt.oldNextV[j]; oNA = a.bt.oldNextADV[j];
          Qf = a.hasAdv ? a.bt.newQ[j] : Vf;
        }
      }
#pragma unroll
      for (int q = 0; q < AGG_N; ++q) ag[q] = g[q];
      if (farOn) {      // this episode's term, updated
        const int pos = a.bt.pos[b], t = pos / farPer, idx = (pos - t * farPer) * 256 + t;
        a.rp.farP[idx] = g[AGG_FRACFAR];
        if (farLds) sFarP[idx] = g[AGG_FRACFAR];
      }
      myMaxAbs = fmaxf(g[AGG_MAXABSERR], 0.f);
      }
      // one LDS atomic per wavefront instead of one per episode (same-address LDS atomics serialise)
      for (int o = 32; o > 0; o >>= 1) myMaxAbs = fmaxf(myMaxAbs, __shfl_xor(myMaxAbs, o, 64));
      if ((tid & 63) == 0) { if (chunkMode == 1) atomicMax(&sc->maxAbsScratch, __float_as_uint(myMaxAbs)); else atomicMax(sMaxAbs, __float_as_uint(myMaxAbs)); }
    }
    if (chunkMode == 1) return;
    if (chunkMode == 2 && tid == 0) { *sMaxAbs = sc->maxAbsScratch; sc->maxAbsScratch = 0u; }
    PSTAMP(sc, 18);
    __syncthreads();
    PSTAMP(sc, 19);
    unsigned long long* sScan = reinterpret_cast<unsigned long long*>(sFarDelta) + 2;
    const int farCnt = farSegment(nEp, farPer);
    const float* const gFarP = a.rp.farP; const float* const gFarN = a.rp.farN;     // (locals: a lambda capturing `a` would put the whole record on the stack)
    unsigned long long* const gFarStart = a.rp.farStart;
    long long farTotal = nFarTot0;
    if (defer) {
    } else if (farLds) {
#pragma unroll
      for (int i = 0; i < FAR_REGS; ++i) farT[i] = sFarP[i * 256 + tid];
      TSTAMP(sc, 14);
      unsigned long long tot = 0;
      // a count or a term outside the range of the float recurrence: the emulated loop over the LDS copy
      if (!farCountRegs(farT, farL, farG0, gFarStart, reinterpret_cast<unsigned*>(sScan), &tot)) tot = farCountMem<false>(sFarP, gFarN, farCnt, gFarStart, sScan);
#ifdef HL_TAIL_STAMPS
      if (tid == 0 && (long long)(tot >> 48) > sc->dbgT[12]) sc->dbgT[12] = (long long)(tot >> 48);
      tot &= 0xffffffffffffull;
#endif
      farTotal = (long long)tot;
      TSTAMP(sc, 15);
    } else {
      __threadfence();
      farTotal = (long long)farCountMem<true>(gFarP, gFarN, farCnt, gFarStart, sScan);
    }
    if (tid == 0) {
      long long nFarTot = farTotal;
      maxAll = fmaxf(maxAll0, __uint_as_float(*sMaxAbs));
      // updateTrainingStatistics: ReF-ER clip annealing for the NEXT sampling (:193-196)
      const double Cm = 1 + a.clipImpWeight / (1 + (double)(nGrad0 + 1) * a.epsAnneal);
      if (Cm <= 1) nFarTot = 0;
      nFarStat = nFarTot;
      sc->maxAbsErrAll = maxAll; sc->maxAbsErrStep = maxAll; sc->Cmax = Cm; sc->Cinv = 1 / Cm; sc->cnt[3] = nTrans;
      if (!defer) { sc->nFarTotal = nFarTot; sc->nFarStat = nFarStat; sc->cnt[2] = nFarStat; }
      if (a.nRanks > 1) { sc->cnt[0] = sc->seenLocal[0]; sc->cnt[1] = sc->seenLocal[1]; }   // undo the last all-reduce
      if (a.cntMsg) {       // counters ride in the tail of the gradient buffer: one all-reduce per step instead of two
        const long long c4[4] = {sc->seenLocal[0], sc->seenLocal[1], nFarStat, nTrans};
        encodeCounters(a.cntMsg, c4);
      }
    }
  }
  if ((mode & POST_ENCODE) && a.cntMsg && tid == 0) {   // eager steps: the counters as they stand after the removal pass
    const long long c4[4] = {sc->seenLocal[0], sc->seenLocal[1], sc->cnt[2], sc->cnt[3]};
    encodeCounters(a.cntMsg, c4);
  }
  if ((mode & (POST_BETA | POST_INIT)) && tid == 0) {
    // updateCounters (:46-92); with several replicas cnt[] holds the all-reduced counters
    const bool rewritten = (mode & POST_AGG) && a.nRanks > 1;      // (the bookkeeping part above put the local counters back)
    long long cntR[4] = {rewritten ? sc->cnt[0] : cnt0, rewritten ? sc->cnt[1] : cnt1, cnt2, cnt3};
    if (a.cntMsg && (mode & POST_BETA)) {         // decode the summed chunks (each sum < 2^24: exact)
      for (int c = 0; c < 4; ++c) {
        long long v = 0;
        for (int q = 0; q < 4; ++q) v += (long long)(a.cntMsg[4 * c + q] + 0.5f) << (16 * q);
        cntR[c] = v; sc->cnt[c] = v;
      }
    }
    sc->seenUpd[0] = cntR[0]; sc->seenUpd[1] = cntR[1];
    const long long nFar = a.nRanks > 1 ? cntR[2] : nFarStat;
    const long long nStored = a.nRanks > 1 ? cntR[3] : nTrans;
    const double fracOffPol = (double)nFar / (double)(nStored > 1 ? nStored : 1);
    const double nDataSize = fmax(a.maxObsGlobal, (double)nStored);
    const double learnRefer = 0.1 * a.batchGlobal / nDataSize;
    if (!defer) refEerPenal(sc, fracOffPol, learnRefer, a.penalTol, beta0, alpha0, false);
    if (mode & POST_BETA) {
      // stats.maxAbsError EMA (:239-240) uses the replica-local data size
      const double lrLoc = 0.1 * a.batchGlobal / fmax(a.maxObsGlobal, (double)nTrans);
      sc->maxAbsErrEMA = ema0 + lrLoc * ((double)maxAll - ema0);
      double nb1 = bt1 * 0.9; if (nb1 < (double)FLT_EPSILON) nb1 = 0;
      double nb2 = bt2 * 0.999; if (nb2 < (double)FLT_EPSILON) nb2 = 0;
      sc->adam_bt1 = nb1; sc->adam_bt2 = nb2;
      sc->nStep = nStep0 + 1;
      sc->nGradSteps = nGrad0 + 1;
      // Adam step size of the NEXT step (read by its dW epilogue from the other buffer slot)
      sc->etaEff[a.parity ^ 1] = adamEtaEff(nStep0 + 1, nb1, nb2, a.eta0, a.epsAnneal);
    }
  }
  PSTAMP(sc, 20);
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
__device__ __forceinline__ int scanRows(int K, const bool* flag, int* excl, int* sWave /*[4]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
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

// sampler phases: A = draw + Lemire acceptance, B = sort / unique / redraw + Adam draws,
// C = index -> (episode, step), truncated-next rows, gather.  Between phases the candidate
// indices live in bt.sVals (HBM) and the generator in DevScalars::rng.

// draw accepted values into vals[from..B) (Lemire rejection, words consumed in order)
__device__ void drawAccepted(unsigned* x, unsigned* xo, int* sPos, unsigned* raw, unsigned* vals, int* sWave,
                             int from, int B, int K, unsigned range, unsigned threshold) {
  const int tid = threadIdx.x;
  int filled = from;
  while (filled < B) {
    const int need = B - filled;
    mtDraw(x, xo, sPos, raw, need);
    bool fl[SMAXK]; int ex[SMAXK]; unsigned v[SMAXK];
    _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
      const int i = r * 256 + tid;
      fl[r] = false; v[r] = 0;
      if (i < need) {
        const unsigned long long prod = (unsigned long long)raw[i] * (unsigned long long)range;
        fl[r] = (unsigned)prod >= threshold; v[r] = (unsigned)(prod >> 32);
      }
    }
    const int acc = scanRows(K, fl, ex, sWave);
    _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) if (fl[r]) vals[filled + ex[r]] = v[r];
    filled += acc;
  }
  __syncthreads();
}
// TSample_impRank / TSample_impErr / Sample_impSeq::sample (Sampling.cpp:151-170, 207-230, 270-294): values for vals[from..B)
// through std::discrete_distribution -- generate_canonical<double, 53> from two generator words, lower_bound over the
// cumulative table -- and, for PERseq, a step drawn with uniform_real_distribution<float> (one more word) times the episode's
// length.  No rejection: every value consumes its two (three) words in order.
__device__ __forceinline__ void drawPER(const SampleArgs& a, unsigned* x, unsigned* xo, int* sPos, unsigned* raw, unsigned* vals, int from, int B) {
  const int tid = threadIdx.x, Wn = a.perAlgo == HL_SAMPLE_PERSEQ ? 3 : 2;
  for (int c0 = from; c0 < B; c0 += 256) {
    const int n = min(256, B - c0);
    mtDraw(x, xo, sPos, raw, n * Wn);
    if (tid < n) {
      const unsigned w0 = raw[Wn * tid], w1 = raw[Wn * tid + 1];
      double p = ((double)w0 + (double)w1 * 4294967296.0) / 18446744073709551616.0;
      if (p >= 1.0) p = 0.99999999999999988898;                  // nextafter(1, 0)
      long long lo = 0;
      if (a.perN >= 2) {                                          // std::lower_bound: first entry not less than p
        long long len = a.perN;
        while (len > 0) { const long long half = len >> 1; if (a.perCp[lo + half] < p) { lo += half + 1; len -= half + 1; } else len = half; }
      }
      unsigned v = (unsigned)lo;
      if (a.perAlgo == HL_SAMPLE_PERSEQ) {
        float u = (float)raw[Wn * tid + 2] / 4294967296.0f;
        if (u >= 1.0f) u = 0.99999994f;                           // nextafterf(1, 0)
        const PosRec rec = a.rp.posRec[lo];
        v = (unsigned)(rec.prefix + (long long)(unsigned long long)(u * (float)(unsigned long long)(rec.N - 1)));
      }
      vals[c0 + tid] = v;
    }
    __syncthreads();
  }
}
// sort vals[0..B) and remove duplicates (std::sort + std::unique); returns the unique count
__device__ int sortUnique(unsigned* vals, int* sWave, int B, int Bp, int K) {
  const int tid = threadIdx.x;
  for (int i = B + tid; i < Bp; i += 256) vals[i] = 0xFFFFFFFFu;
  bitonicSort(vals, Bp);
  bool fl[SMAXK]; int ex[SMAXK]; unsigned v[SMAXK];
  _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
    const int i = r * 256 + tid;
    v[r] = i < B ? vals[i] : 0u;
    fl[r] = i < B && (i == 0 || v[r] != vals[i - 1]);
  }
  const int nu = scanRows(K, fl, ex, sWave);     // (barriers inside: all reads of vals are done)
  _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) if (fl[r]) vals[ex[r]] = v[r];
  __syncthreads();
  return nu;
}

// LDS footprint of the tail code (carved from the hosting kernel's LDS block, which the extra
// workgroup does not otherwise use)
#define TAIL_LDS_BYTES (8 * SMAXB + 4 * (624 + 624 + SMAXB + SMAXB + SMAXB) + 64)

// IDtoSeqStep for one flat index: interpolation guess into the per-position table, then binary search (Sampling.cpp:26-47)
__device__ __forceinline__ PosRec findPosition(const DevReplay& rp, long long f, int nEp, unsigned long long nData, int* pos) {
  int k0 = (int)(((double)f * (double)nEp) / (double)nData);
  k0 = min(max(k0, 0), nEp - 1);
  PosRec rec = rp.posRec[k0];
  const long long p1 = rp.posRec[k0 + 1].prefix;
  int lo = k0;
  if (f < rec.prefix || f >= p1) {
    int l = f < rec.prefix ? 0 : k0 + 1, hgh = f < rec.prefix ? k0 : nEp;   // largest k in [l,hgh): prefix[k] <= f
    while (hgh - l > 1) { const int mid = (l + hgh) >> 1; if (rp.posRec[mid].prefix <= f) l = mid; else hgh = mid; }
    lo = l; rec = rp.posRec[lo];
  }
  *pos = lo;
  return rec;
}

__device__ __forceinline__ void samplePhases(const SampleArgs& a, int phases, unsigned char* smem) {
  long long* sSlot = reinterpret_cast<long long*>(smem);            // [SMAXB]   (8-byte aligned first)
  unsigned* x = reinterpret_cast<unsigned*>(smem + 8 * SMAXB);      // [624]
  unsigned* xo = x + 624;                                           // [624]
  unsigned* raw = xo + 624;                                         // [SMAXB]
  unsigned* vals = raw + SMAXB;                                     // [SMAXB]
  int* sNextRow = reinterpret_cast<int*>(vals + SMAXB);             // [SMAXB]
  int* sWave = sNextRow + SMAXB;                                    // [4]
  int* sPos = sWave + 4;
  const int tid = threadIdx.x;
  DevScalars* sc = a.sc;
  TSTAMP(sc, 0);
  const int B = a.B;
  // sort network of nextpow2(B) elements, 64 (one wavefront, no barriers) at least: at a local batch of 32 -- one Humanoid
  // replica of eight -- the 256-element network was the longest chain of the launch this rider sits in
  int Bp = 64; while (Bp < B) Bp <<= 1;
  const int K = (Bp + 255) / 256;
  const bool rngNeeded = (phases & (PH_A | PH_B)) != 0;
  if (rngNeeded) {
    const bool bak = a.backupRng && (phases & PH_A);
    for (int k = tid; k < 624; k += 256) { const unsigned v = sc->rng[k]; x[k] = v; if (bak) sc->rngBak[k] = v; }
    if (tid == 0) { const unsigned p0 = sc->rngPos; *sPos = (int)p0; if (bak) sc->rngBakPos = p0; if (phases & PH_A) sc->sampleSeq += 1; }
  }
  const unsigned long long nData = (unsigned long long)sc->nTransitions;
  const int nEp = (int)sc->nEpisodes;
  const unsigned range = (unsigned)nData;
  const unsigned threshold = range ? (0u - range) % range : 0u;
  if ((phases & (PH_B | PH_C)) && !(phases & PH_A))
    for (int i = tid; i < B; i += 256) vals[i] = a.bt.sVals[i];
  __syncthreads();
  TSTAMP(sc, 1);

  if (phases & PH_A) {
    if (a.flatGiven) { for (int i = tid; i < B; i += 256) vals[i] = (unsigned)a.flatGiven[i]; __syncthreads(); }
    else if (a.perAlgo) drawPER(a, x, xo, sPos, raw, vals, 0, B);
    else drawAccepted(x, xo, sPos, raw, vals, sWave, 0, B, K, range, threshold);
  }
  TSTAMP(sc, 2);
  if (phases & PH_B) {
    if (!a.flatGiven) {
      int have = sortUnique(vals, sWave, B, Bp, K);
      TSTAMP(sc, 3);
      while (have < B) {                       // duplicates: redraw the tail (Sampling.cpp:86-93)
        if (a.perAlgo) drawPER(a, x, xo, sPos, raw, vals, have, B);
        else drawAccepted(x, xo, sPos, raw, vals, sWave, have, B, K, range, threshold);
        have = sortUnique(vals, sWave, B, Bp, K);
      }
    }
    // the generator draws of AdamOptimizer::apply_update (one per reference thread)
    if (a.adamDraws > 0) mtDraw(x, xo, sPos, raw, a.adamDraws);
  }
  TSTAMP(sc, 4);
  if (rngNeeded) {
    for (int k = tid; k < 624; k += 256) sc->rng[k] = x[k];
    if (tid == 0) sc->rngPos = (unsigned)*sPos;
  }
  if (!(phases & PH_C)) {
    for (int i = tid; i < B; i += 256) a.bt.sVals[i] = vals[i];
    return;
  }

  // ---- IDtoSeqStep: interpolation guess into the per-position table, then binary search ----
  bool hasNext[SMAXK]; int nextIdx[SMAXK];
  _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
    const int b = r * 256 + tid;
    hasNext[r] = false;
    if (b < B) {
      const long long f = (long long)vals[b];
      int lo;
      const PosRec rec = findPosition(a.rp, f, nEp, nData, &lo);
      const int t = (int)(f - rec.prefix);
      const int e = rec.eidTerm & 0x7fffffff;
      const bool term = rec.eidTerm < 0;
      a.bt.flat[b] = f; a.bt.pos[b] = lo; a.bt.eid[b] = e; a.bt.t[b] = t; a.bt.tag[b] = rec.tag;
      if ((phases & PH_PUBLISH) && !a.selfSearch) __hip_atomic_store(a.bt.slot + b, rec.off + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.bt.slot[b] = rec.off + t;
      sSlot[b] = rec.off + t;
      hasNext[r] = (t + 2 == rec.N && !term);      // Episode::isTruncated(t+1) (Episode.h:158-161)
    }
  }
  TSTAMP(sc, 5);
  const int nNext = scanRows(K, hasNext, nextIdx, sWave);
  TSTAMP(sc, 6);
  _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
    const int b = r * 256 + tid;
    if (b < B) {
      const int nr = hasNext[r] ? B + nextIdx[r] : -1;
      if (hasNext[r]) a.bt.nextSrc[nextIdx[r]] = b;
      if ((phases & PH_PUBLISH) && !a.selfSearch) __hip_atomic_store(a.bt.nextOf + b, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.bt.nextOf[b] = nr;
      sNextRow[b] = nr;
    }
  }
  if (tid == 0) {
    sc->nNext[a.parity] = nNext; sc->nRows[a.parity] = B + nNext;
    // first step of a launch sequence: derive this step's Adam step size from the canonical scalars
    if (a.computeEta) sc->etaEff[a.parity] = adamEtaEff(sc->nStep, sc->adam_bt1, sc->adam_bt2, a.eta0, a.epsAnneal);
  }
  if (a.noGather) return;      // the states are gathered by stack_gather_kernel (conv.hip)
  if (phases & PH_PUBLISH) {   // the gather is done by the helper workgroups (gatherHelper)
    if (a.selfSearch) return;          // ... which ran the search themselves: nobody waits for these arrays inside this kernel
    __builtin_amdgcn_s_waitcnt(0);     // the agent-scope stores of slot / nextOf are acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_store(sc->gatherFlag + a.parity, a.tagSeq ? -sc->sampleSeq : sc->nStep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TSTAMP(sc, 9);
    return;
  }
  __syncthreads();
  TSTAMP(sc, 7);
  // ---- gather: Episode::standardizedState (Episode.h:172-183) for s_t and truncated s_{t+1} ----
  // all loads of a thread are issued before its first store (one exposed HBM round trip); the
  // per-component mean / scale come from LDS (staged at kernel start into raw[], free by now)
  const int dS = a.dS, total = B * dS;
  float* sMean = reinterpret_cast<float*>(raw);            // [dS]   (dS <= SMAXB / 2)
  float* sScale = sMean + SMAXB / 2;
  for (int i = tid; i < dS; i += 256) { sMean[i] = a.rp.stMean[i]; sScale[i] = a.rp.stScale[i]; }
  __syncthreads();
  constexpr int GU = 9;                                    // elements per thread per round (two rounds at B = 256, dS = 17)
  for (int e0 = tid; e0 < total; e0 += 256 * GU) {
    float sv[GU], sn[GU]; int bb[GU], ii[GU], nr[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = e0 + 256 * u;
      bb[u] = -1; sv[u] = 0.f; sn[u] = 0.f; ii[u] = 0; nr[u] = -1;
      if (e < total) {
        const int b = e / dS; bb[u] = b; ii[u] = e - b * dS;
        const long long sl = sSlot[b];
        sv[u] = a.rp.S[(size_t)sl * dS + ii[u]];
        nr[u] = sNextRow[b];
        if (nr[u] >= 0) sn[u] = a.rp.S[(size_t)(sl + 1) * dS + ii[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) if (bb[u] >= 0) {
      const float mu = sMean[ii[u]], scl = sScale[ii[u]];
      a.X0[(size_t)bb[u] * a.ldX0 + ii[u]] = (sv[u] - mu) * scl;
      if (nr[u] >= 0) a.X0[(size_t)nr[u] * a.ldX0 + ii[u]] = (sn[u] - mu) * scl;
    }
  }
  TSTAMP(sc, 8);
}

// helper workgroup `part` of `nParts`: waits for the sampler's hand-off, then gathers its share of
// the minibatch (Episode::standardizedState, Episode.h:172-183).  A single workgroup can keep only
// a few dozen HBM misses in flight; seven of them gather 256 x 17 floats in one round trip.
__device__ __forceinline__ void gatherHelper(const SampleArgs& a, int part, int nParts, unsigned char* smem) {
  float* sMean = reinterpret_cast<float*>(smem);
  float* sScale = sMean + SMAXB / 2;
  const int tid = threadIdx.x, B = a.B, dS = a.dS;
  DevScalars* sc = a.sc;
  if (part == 0) TSTAMP(sc, 21);
  for (int i = tid; i < dS; i += 256) { sMean[i] = a.rp.stMean[i]; sScale[i] = a.rp.stScale[i]; }
  if (a.selfSearch) {
    // No hand-off: every helper runs the search over the sorted indices itself (the rider's copy of it feeds the bookkeeping
    // arrays meanwhile) -- its publication (stores, their acknowledgement, the flag, the helpers' coherent re-reads: 2 us) was
    // the longest chain of the dW launch.  Same table, same scan: the same slots and next-state rows as the rider finds.
    long long* sSlot = reinterpret_cast<long long*>(smem + 8 * (SMAXB / 2));       // [SMAXB]  (behind mean | scale)
    int* sNextRow = reinterpret_cast<int*>(sSlot + SMAXB);                         // [SMAXB]
    int* sWave = sNextRow + SMAXB;                                                 // [4]
    const unsigned long long nData = (unsigned long long)sc->nTransitions;
    const int nEp = (int)sc->nEpisodes;
    const int K = (B + 255) / 256;
    bool hasNext[SMAXK]; int nextIdx[SMAXK];
    _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) {
      const int b = r * 256 + tid;
      hasNext[r] = false;
      if (b < B) {
        const long long f = (long long)a.bt.sVals[b];
        int lo;
        const PosRec rec = findPosition(a.rp, f, nEp, nData, &lo);
        const int t = (int)(f - rec.prefix);
        sSlot[b] = rec.off + t;
        hasNext[r] = (t + 2 == rec.N && !(rec.eidTerm < 0));
      }
    }
    scanRows(K, hasNext, nextIdx, sWave);
    _Pragma("unroll") for (int r = 0; r < SMAXK; ++r) if (r < K) { const int b = r * 256 + tid; if (b < B) sNextRow[b] = hasNext[r] ? B + nextIdx[r] : -1; }
    __syncthreads();
    if (part == 0) TSTAMP(sc, 22);
    const int per = (B + nParts - 1) / nParts, b0 = part * per, b1 = min(B, b0 + per);
    const int total = max(0, b1 - b0) * dS;
    constexpr int GU = 4;
    for (int e0 = tid; e0 < total; e0 += 256 * GU) {
      float sv[GU], sn[GU]; int bb[GU], ii[GU], nr[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int e = e0 + 256 * u;
        bb[u] = -1; sv[u] = 0.f; sn[u] = 0.f; ii[u] = 0; nr[u] = -1;
        if (e < total) {
          const int b = b0 + e / dS; bb[u] = b; ii[u] = e - (b - b0) * dS;
          const long long sl = sSlot[b];
          sv[u] = a.rp.S[(size_t)sl * dS + ii[u]];
          nr[u] = sNextRow[b];
          if (nr[u] >= 0) sn[u] = a.rp.S[(size_t)(sl + 1) * dS + ii[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) if (bb[u] >= 0) {
        const float mu = sMean[ii[u]], scl = sScale[ii[u]];
        a.X0[(size_t)bb[u] * a.ldX0 + ii[u]] = (sv[u] - mu) * scl;
        if (nr[u] >= 0) a.X0[(size_t)nr[u] * a.ldX0 + ii[u]] = (sn[u] - mu) * scl;
      }
    }
    if (part == 0) TSTAMP(sc, 23);
    return;
  }
  if (tid == 0) {
    const long long want = a.tagSeq ? -sc->sampleSeq : sc->nStep + 1;       // (negative: never equal to a tag of the other kind)
    int spins = 0;
    while (__hip_atomic_load(sc->gatherFlag + a.parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { sc->errFlag = 78; break; }
    }
  }
  __syncthreads();
  if (part == 0) TSTAMP(sc, 22);
  const int per = (B + nParts - 1) / nParts, b0 = part * per, b1 = min(B, b0 + per);
  const int total = max(0, b1 - b0) * dS;
  constexpr int GU = 4;
  for (int e0 = tid; e0 < total; e0 += 256 * GU) {
    float sv[GU], sn[GU]; int bb[GU], ii[GU], nr[GU]; long long sl[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = e0 + 256 * u;
      bb[u] = -1; nr[u] = -1; sl[u] = 0; ii[u] = 0;
      if (e < total) {
        const int b = b0 + e / dS; bb[u] = b; ii[u] = e - (b - b0) * dS;
        sl[u] = __hip_atomic_load(a.bt.slot + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nr[u] = __hip_atomic_load(a.bt.nextOf + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      sv[u] = 0.f; sn[u] = 0.f;
      if (bb[u] >= 0) {
        sv[u] = a.rp.S[(size_t)sl[u] * dS + ii[u]];
        if (nr[u] >= 0) sn[u] = a.rp.S[(size_t)(sl[u] + 1) * dS + ii[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) if (bb[u] >= 0) {
      const float mu = sMean[ii[u]], scl = sScale[ii[u]];
      a.X0[(size_t)bb[u] * a.ldX0 + ii[u]] = (sv[u] - mu) * scl;
      if (nr[u] >= 0) a.X0[(size_t)nr[u] * a.ldX0 + ii[u]] = (sn[u] - mu) * scl;
    }
  }
  if (part == 0) TSTAMP(sc, 23);
}

__device__ __forceinline__ void postPhase(const PostArgs& a, unsigned char* smem) {
  long long* sFarDelta = reinterpret_cast<long long*>(smem);        // [0] unused, [1] max |error| bits, [2..5] scan scratch of farExact
  unsigned* sMaxAbs = reinterpret_cast<unsigned*>(smem + 8);
  postPart(a, sFarDelta, sMaxAbs, reinterpret_cast<float*>(smem + 64), (TAIL_LDS_BYTES - 64) / 4);
}

// What a POST_DEFER bookkeeping pass left over, as a rider (block 1) of the NEXT step's fused kernel: the far-policy count over
// the fractions that pass patched (dev_common.h) and the beta / alpha update that needs it.  The heads of this kernel read
// beta late (fused.hip): they wait for betaSeq, published here with agent-scope stores (the first acknowledged before the second
// is issued; no cache-wide release, which costs tens of microseconds).  256 threads.
__device__ __forceinline__ void farBetaPhase(const PostArgs& a, unsigned char* smem) {
  DevScalars* sc = a.sc;
  const int tid = threadIdx.x;
  TSTAMP(sc, 10);
  const double Cmax = sc->Cmax, beta0 = sc->beta, alpha0 = sc->alpha;
  const long long nTrans = sc->nTransitions, nEpL = sc->nEpisodes, nGrad = sc->nGradSteps;
  const int nEp = (int)nEpL, per = (nEp + 255) / 256;
  unsigned long long* sScan = reinterpret_cast<unsigned long long*>(smem);
  const float* const gFarP = a.rp.farP; const float* const gFarN = a.rp.farN; unsigned long long* const gFarStart = a.rp.farStart;
  unsigned long long tot = 0;
  bool done = false;
  if (per <= FAR_REGS) {
    float f[FAR_REGS], l[FAR_REGS]; unsigned long long g0[FAR_SUB];
#pragma unroll
    for (int c = 0; c < FAR_SUB; ++c) g0[c] = gFarStart[c * 256 + tid];
#pragma unroll
    for (int i = 0; i < FAR_REGS; ++i) { f[i] = gFarP[i * 256 + tid]; l[i] = gFarN[i * 256 + tid]; }
    done = farCountRegs(f, l, g0, gFarStart, reinterpret_cast<unsigned*>(sScan), &tot);
#ifdef HL_TAIL_STAMPS
    if (tid == 0 && (long long)(tot >> 48) > sc->dbgT[12]) sc->dbgT[12] = (long long)(tot >> 48);      // most rounds seen in any step (tools/far_rounds.py)
    tot &= 0xffffffffffffull;
#endif
  }
  if (!done) tot = farCountMem<false>(gFarP, gFarN, farSegment(nEp, per), gFarStart, sScan);
  TSTAMP(sc, 11);
  if (tid != 0) return;
  const long long nFar = Cmax <= 1 ? 0 : (long long)tot;
  sc->nFarTotal = nFar; sc->nFarStat = nFar; sc->cnt[2] = nFar;
  const double fracOffPol = (double)nFar / (double)(nTrans > 1 ? nTrans : 1);
  const double learnRefer = 0.1 * a.batchGlobal / fmax(a.maxObsGlobal, (double)nTrans);
  refEerPenal(sc, fracOffPol, learnRefer, a.penalTol, beta0, alpha0, true);
  __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): beta and alpha are in memory
  __hip_atomic_store(&sc->betaSeq, nGrad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the extra workgroup of an MLP kernel: role 1 = sampler phases (for the NEXT step), 2 = bookkeeping, 3 = far-policy count + beta
__device__ __forceinline__ void runExtra(const ExtraArgs& ex, unsigned char* smem) {
  if (ex.role == 1) samplePhases(ex.samp, ex.phases, smem);
  else if (ex.role == 2) postPhase(ex.post, smem);
  else if (ex.role == 3) farBetaPhase(ex.post, smem);      // what a POST_DEFER bookkeeping pass of an earlier launch of this step left over
}

}  // namespace hl
