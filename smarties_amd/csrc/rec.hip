// smarties_amd/csrc/rec.hip -- recurrent (LSTM) hidden layers with truncated back-propagation through time.
//
//   reference: Network/Layers/Layer_LSTM.h:78-165 (forward / backward of one step), Network/Network.h:102-193
//   (forward with the previous step as recurrent input, backProp over the time series), Approximator.h:116-173
//   (every step of the window is forwarded), ReplayMemory/MemoryBuffer.cpp:391-402 (the window: min(nnBPTTseq, t)
//   steps before the sampled one), Network/Layers/Layers.h:324-393 (parametric residual).
//
// First device version of this path: ONE workgroup per sample walks the sample's window step by step (the recurrence
// is sequential; samples are independent), gates one per thread, weights read through the L2.  It stores, per
// (sample, step) row, the operands of the weight-gradient contractions -- inputs [in | previous output] and the four
// gate deltas -- so that all weight gradients (and Adam) are formed by the same dW kernel as for dense layers, as
// X^T delta over the rows; rows of unused steps carry zero deltas.
#include "tail_dev.h"

namespace hl {

#define REC_MAXC 64       // cells per layer (4 gates x 64 = 256 threads)
#define REC_MAXIN 256     // inputs of the first layer
#define REC_STATES 4608   // window states kept in LDS (e.g. 18 steps x 256 state components)

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the global stores of the step
// (the rows kept for the backward pass / the dW launch), ~1 us each, and nothing in these kernels reads them back
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float recSigm(float in) {     // Sigm::_eval (Functions.h:158-165), safeExp cut at 8 (Definitions.h:43)
  if (in > 0.f) return 1.f / (1.f + expf(fminf(8.f, fmaxf(-8.f, -in))));
  const float ex = expf(fminf(8.f, fmaxf(-8.f, in)));
  return ex / (1.f + ex);
}

// weights of all LSTM layers staged in LDS with a padded row stride (4 nC + 1: the forward pass reads columns, the backward
// pass rows, both conflict-free); nets that do not fit read them through the L2 (ldsW = 0)
__device__ __forceinline__ void recStageWeights(const RecArgs& a, float* sW, int tid) {
  int off = 0;
  for (int j = 0; j < a.nL; ++j) {
    const RecLayer& L = a.L[j];
    const int NO = a.gates * L.nC, rows = L.nIn + L.nC;
    const float* src = a.W + L.indW;
    // eight loads in flight per thread (a rolled loop pays one L2 / HBM round trip per element: 50 in a row at 32 cells)
    const int total = rows * NO;
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; v[u] = e < total ? src[e] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < total) { const int i = e / NO, o = e - i * NO; sW[off + i * (NO + 1) + o] = v[u]; } }
    }
    off += rows * (NO + 1);
  }
}
// offset of layer j's weights inside the LDS copy
__device__ __forceinline__ int recLdsOffset(const RecArgs& a, int j) {
  int off = 0;
  for (int q = 0; q < j; ++q) off += (a.L[q].nIn + a.L[q].nC) * (a.gates * a.L[q].nC + 1);
  return off;
}

template <bool LDSW>
__global__ __launch_bounds__(256) void rec_forward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sBuf[2][REC_MAXIN];                    // input of the current layer / output of the current block
  __shared__ float sPrevOut[HL_MAX_HIDDEN][REC_MAXC], sPrevSt[HL_MAX_HIDDEN][REC_MAXC];
  __shared__ float sX[4 * REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  // acting (MemoryBuffer::agentToMinibatch, MemoryBuffer.cpp:440-467): the agent's last steps, from a zero recurrent state
  const bool acting = a.actStates != nullptr;
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  // layer descriptors copied out of the kernel-argument segment once (inside the step loops every field access was a scalar
  // load of its own)
  RecLayer LL[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float bias[HL_MAX_HIDDEN], wr[HL_MAX_HIDDEN], br[HL_MAX_HIDDEN];      // this thread's gate bias / residual parameters per layer
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) {
    bias[j] = 0.f; wr[j] = 0.f; br[j] = 0.f;
    if (j < a.nL) {
      const RecLayer& L = a.L[j];
      if (tid < 4 * L.nC) bias[j] = W[L.indB + tid];
      if (L.hasRes && tid < L.resW) { wr[j] = W[L.indWr + tid]; br[j] = W[L.indBr + tid]; }
    }
  }
  // the standardised states of the whole window, fetched in one round (Episode::standardizedState, Episode.h:172-183)
  __shared__ float sStates[REC_STATES];
  const bool preload = nSteps * a.dS <= REC_STATES;
  if (preload) for (int e = tid; e < nSteps * a.dS; e += 256) {
    const int kk = e / a.dS, i = e - kk * a.dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * a.dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  ldsBarrier();
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    const long long sl = slot - T + k;
    if (tid < a.dS) {
      if (preload) sBuf[0][tid] = sStates[k * a.dS + tid];
      else { const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)sl * a.dS + tid]; sBuf[0][tid] = (raw - sMean) * sScale; }
    }
    ldsBarrier();
    int cur = 0;
#pragma unroll
    for (int j = 0; j < HL_MAX_HIDDEN; ++j) if (j < a.nL) {
      const RecLayer& L = LL[j];
      const int nIn = L.nIn, nC = L.nC, NO = 4 * nC;
      const float* in = sBuf[cur];
      // element (i, o) of [W_in; W_rec] of this layer: LDS copy (padded rows) or global memory, decided at compile time
      const int ldw = LDSW ? NO + 1 : NO, wOff = LDSW ? recLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * ldw + o]; else return gWj[(size_t)i * ldw + o]; };
      if (store) {
        for (int i = tid; i < nIn; i += 256) L.A[r * L.ldA + i] = in[i];
        if (tid < nC) L.A[r * L.ldA + nIn + tid] = k > 0 ? sPrevOut[j][tid] : 0.f;
      }
      if (tid < NO) {
        float acc = bias[j];
        // (unrolled: the LDS reads of eight terms are in flight together; a rolled loop pays the LDS latency per term)
#pragma unroll 8
        for (int i = 0; i < nIn; ++i) acc += in[i] * wAt(i, tid);
        if (k > 0) {
#pragma unroll 8
          for (int i = 0; i < nC; ++i) acc += sPrevOut[j][i] * wAt(nIn + i, tid);
        }
        if (tid >= nC) acc = recSigm(acc);                 // the gates overwrite their inputs
        sX[tid] = acc;
        if (store) L.X[r * NO + tid] = acc;
      }
      ldsBarrier();
      float out = 0.f, st = 0.f;
      if (tid < nC) {
        st = sX[tid] * sX[nC + tid] + (k > 0 ? sPrevSt[j][tid] * sX[2 * nC + tid] : 0.f);
        const float co = actEval(HL_FUNC_TANH, st);
        out = sX[3 * nC + tid] * co;
        if (store) { L.Y[r * NO + tid] = out; L.Y[r * NO + nC + tid] = st; L.Y[r * NO + 2 * nC + tid] = co; }
        float blk = out;                                   // ParametricResidualLayer::forward (Layers.h:347-361)
        if (L.hasRes && tid < L.resW) blk += in[tid] * wr[j] + br[j];
        sBuf[cur ^ 1][tid] = blk;
      }
      ldsBarrier();
      if (tid < nC) { sPrevOut[j][tid] = out; sPrevSt[j][tid] = st; }
      cur ^= 1;
    }
    const int nCl = a.L[a.nL - 1].nC;
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    ldsBarrier();
  }
}

template <bool LDSW>
__global__ __launch_bounds__(256) void rec_backward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_MAXIN];                    // error w.r.t. the output of the current block (from above, same step)
  __shared__ float sRec[HL_MAX_HIDDEN][REC_MAXC];          // error w.r.t. this step's LSTM output coming from step k+1
  __shared__ float sNxtSt[HL_MAX_HIDDEN][REC_MAXC], sNxtF[HL_MAX_HIDDEN][REC_MAXC];
  __shared__ float sD[4 * REC_MAXC], sRes[REC_MAXC], sRecNew[REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  // layer descriptors copied out of the kernel-argument segment once (inside the step loops every field access was a scalar
  // load of its own)
  RecLayer LL[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float wr[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) { wr[j] = 0.f; if (j < a.nL && a.L[j].hasRes && tid < a.L[j].resW) wr[j] = W[a.L[j].indWr + tid]; }
  ldsBarrier();
  // rows of the steps this sample does not have: zero deltas (their stale inputs then add nothing to the gradients)
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      if (tid < 4 * L.nC) L.D[r * 4 * L.nC + tid] = 0.f;
      if (L.hasRes && tid < L.nC) L.Rd[r * L.ldR + tid] = 0.f;
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    const int nCl = a.L[a.nL - 1].nC;
    if (tid < nCl) sTop[0][tid] = k == T ? a.Dres[(size_t)b * a.ldD + tid] : 0.f;
    // this step's stored activations of every layer, fetched in one round
    float vCo[HL_MAX_HIDDEN], vCi[HL_MAX_HIDDEN], vIG[HL_MAX_HIDDEN], vFG[HL_MAX_HIDDEN], vOG[HL_MAX_HIDDEN], vPs[HL_MAX_HIDDEN];
#pragma unroll
    for (int j = 0; j < HL_MAX_HIDDEN; ++j) {
      vCo[j] = vCi[j] = vIG[j] = vFG[j] = vOG[j] = vPs[j] = 0.f;
      if (j < a.nL && tid < a.L[j].nC) {
        const RecLayer& L = a.L[j]; const int nC = L.nC, NO = 4 * nC;
        vCo[j] = L.Y[r * NO + 2 * nC + tid]; vCi[j] = L.X[r * NO + tid]; vIG[j] = L.X[r * NO + nC + tid];
        vFG[j] = L.X[r * NO + 2 * nC + tid]; vOG[j] = L.X[r * NO + 3 * nC + tid];
        if (k > 0) vPs[j] = L.Y[(r - 1) * NO + nC + tid];
      }
    }
    ldsBarrier();
#pragma unroll
    for (int j = HL_MAX_HIDDEN - 1; j >= 0; --j) if (j < a.nL) {
      const RecLayer& L = LL[j];
      const int nIn = L.nIn, nC = L.nC, NO = 4 * nC;
      const int ldw = LDSW ? NO + 1 : NO, wOff = LDSW ? recLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * ldw + o]; else return gWj[(size_t)i * ldw + o]; };
      if (tid < nC) {
        const float eTop = sTop[cur][tid];
        // ParametricResidualLayer::backward (Layers.h:363-393): the delta passes to the LSTM output, and through w to the block input
        if (L.hasRes) { L.Rd[r * L.ldR + tid] = eTop; sRes[tid] = tid < L.resW ? eTop * wr[j] : 0.f; }
        const float D = eTop + (k < T ? sRec[j][tid] : 0.f);
        // LSTMLayer::backward (Layer_LSTM.h:127-165)
        const float co = vCo[j];
        const float cellInpt = vCi[j], IG = vIG[j], FG = vFG[j], OG = vOG[j];
        const float diff = (1.f - co * co) * D;
        const float sd = diff * OG + (k < T ? sNxtSt[j][tid] * sNxtF[j][tid] : 0.f);
        const float d0 = IG * sd;
        const float d1 = IG * (1.f - IG) * cellInpt * sd;
        const float d2 = k > 0 ? FG * (1.f - FG) * vPs[j] * sd : 0.f;
        const float d3 = OG * (1.f - OG) * D * co;
        sD[tid] = d0; sD[nC + tid] = d1; sD[2 * nC + tid] = d2; sD[3 * nC + tid] = d3;
        L.D[r * NO + tid] = d0; L.D[r * NO + nC + tid] = d1; L.D[r * NO + 2 * nC + tid] = d2; L.D[r * NO + 3 * nC + tid] = d3;
        sNxtSt[j][tid] = sd; sNxtF[j][tid] = FG;
      }
      ldsBarrier();
      // Layer::backward (Layers.h:123-188): errors to the block below (not below the first layer) and to the previous step
      // one row of [W_in; W_rec] per group of four lanes (quarter sums joined by two shuffles): rows 0..nIn-1 give the error
      // of the block below (skipped under the first layer), rows nIn.. the error handed to the previous step
      {
        const int part = tid & 3, row0 = j > 0 ? 0 : nIn, nRow = nIn + (k > 0 ? nC : 0);
        for (int i0 = row0; i0 < nRow; i0 += 64) {
          const int i = i0 + (tid >> 2);
          float e = 0.f;
          if (i < nRow) {
#pragma unroll 8
            for (int u = 0; u < NO / 4; ++u) { const int o = 4 * u + part; e += wAt(i, o) * sD[o]; }
          }
          e += __shfl_xor(e, 1, 64); e += __shfl_xor(e, 2, 64);
          if (part == 0 && i < nRow) {
            if (i < nIn) sTop[cur ^ 1][i] = (L.hasRes && i < L.resW ? sRes[i] : 0.f) + e;
            else sRecNew[i - nIn] = e;
          }
        }
      }
      ldsBarrier();
      if (tid < nC) sRec[j][tid] = k > 0 ? sRecNew[tid] : 0.f;
      cur ^= 1;
    }
    ldsBarrier();
  }
}

// ---- MGU layers (Network/Layers/Layer_GRU.h): forget = sigm(Wff in + Wfr prevOut + bf), state = tanh(Wsf in + Wsr (forget * prevOut)
// + bs), output = forget * state + (1 - forget) * prevOut.  Same structure as the LSTM kernels: one workgroup per sample. ----
template <bool LDSW>
__global__ __launch_bounds__(256) void mgu_forward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sBuf[2][REC_MAXIN];
  __shared__ float sPrevOut[HL_MAX_HIDDEN][REC_MAXC];
  __shared__ float sF[REC_MAXC], sS[REC_MAXC];
  __shared__ float sStates[REC_STATES];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool acting = a.actStates != nullptr;
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  // layer descriptors copied out of the kernel-argument segment once (inside the step loops every field access was a scalar
  // load of its own)
  RecLayer LL[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float bias[HL_MAX_HIDDEN], wr[HL_MAX_HIDDEN], br[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) {
    bias[j] = 0.f; wr[j] = 0.f; br[j] = 0.f;
    if (j < a.nL) {
      const RecLayer& L = a.L[j];
      if (tid < 2 * L.nC) bias[j] = W[L.indB + tid];
      if (L.hasRes && tid < L.resW) { wr[j] = W[L.indWr + tid]; br[j] = W[L.indBr + tid]; }
    }
  }
  const bool preload = nSteps * a.dS <= REC_STATES;
  if (preload) for (int e = tid; e < nSteps * a.dS; e += 256) {
    const int kk = e / a.dS, i = e - kk * a.dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * a.dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  ldsBarrier();
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    if (tid < a.dS) {
      if (preload) sBuf[0][tid] = sStates[k * a.dS + tid];
      else { const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)(slot - T + k) * a.dS + tid]; sBuf[0][tid] = (raw - sMean) * sScale; }
    }
    ldsBarrier();
    int cur = 0;
#pragma unroll
    for (int j = 0; j < HL_MAX_HIDDEN; ++j) if (j < a.nL) {
      const RecLayer& L = LL[j];
      const int nIn = L.nIn, nC = L.nC, NO = 2 * nC;
      const float* in = sBuf[cur];
      const int ldw = LDSW ? NO + 1 : NO, wOff = LDSW ? recLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * ldw + o]; else return gWj[(size_t)i * ldw + o]; };
      if (store) {
        for (int i = tid; i < nIn; i += 256) L.A[r * L.ldA + i] = in[i];
        if (tid < nC) L.A[r * L.ldA + nIn + tid] = k > 0 ? sPrevOut[j][tid] : 0.f;
      }
      float acc = 0.f;
      if (tid < NO) {
        acc = bias[j];
#pragma unroll 8
        for (int i = 0; i < nIn; ++i) acc += in[i] * wAt(i, tid);
        if (tid < nC) {          // forget gate
          if (k > 0) {
#pragma unroll 8
            for (int i = 0; i < nC; ++i) acc += wAt(nIn + i, tid) * sPrevOut[j][i];
          }
          acc = recSigm(acc);
          sF[tid] = acc;
          if (store) L.X[r * NO + tid] = acc;
        }
      }
      ldsBarrier();
      if (tid >= nC && tid < NO) {   // cell state
        if (k > 0) {
#pragma unroll 8
          for (int i = 0; i < nC; ++i) acc += wAt(nIn + i, tid) * sPrevOut[j][i] * sF[i];
        }
        acc = actEval(HL_FUNC_TANH, acc);
        sS[tid - nC] = acc;
        if (store) L.X[r * NO + tid] = acc;
      }
      ldsBarrier();
      float out = 0.f;
      if (tid < nC) {
        const float f = sF[tid], st = sS[tid], po = k > 0 ? sPrevOut[j][tid] : 0.f;
        out = k > 0 ? f * st + (1.f - f) * po : f * st;
        if (store) { L.Y[r * NO + tid] = out; L.A2[r * L.ldA2 + tid] = po * f; }
        float blk = out;
        if (L.hasRes && tid < L.resW) blk += in[tid] * wr[j] + br[j];
        sBuf[cur ^ 1][tid] = blk;
      }
      ldsBarrier();
      if (tid < nC) sPrevOut[j][tid] = out;
      cur ^= 1;
    }
    const int nCl = a.L[a.nL - 1].nC;
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    ldsBarrier();
  }
}

template <bool LDSW>
__global__ __launch_bounds__(256) void mgu_backward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_MAXIN];
  __shared__ float sRec[HL_MAX_HIDDEN][REC_MAXC];          // dLdprevOut handed from step k+1 to step k
  __shared__ float sDF[REC_MAXC], sDS[REC_MAXC], sFP[REC_MAXC], sRes[REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  // layer descriptors copied out of the kernel-argument segment once (inside the step loops every field access was a scalar
  // load of its own)
  RecLayer LL[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float wr[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) { wr[j] = 0.f; if (j < a.nL && a.L[j].hasRes && tid < a.L[j].resW) wr[j] = W[a.L[j].indWr + tid]; }
  ldsBarrier();
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      if (tid < 2 * L.nC) L.D[r * 2 * L.nC + tid] = 0.f;
      if (L.hasRes && tid < L.nC) L.Rd[r * L.ldR + tid] = 0.f;
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    const int nCl = a.L[a.nL - 1].nC;
    if (tid < nCl) sTop[0][tid] = k == T ? a.Dres[(size_t)b * a.ldD + tid] : 0.f;
    float vF[HL_MAX_HIDDEN], vS[HL_MAX_HIDDEN], vP[HL_MAX_HIDDEN];
#pragma unroll
    for (int j = 0; j < HL_MAX_HIDDEN; ++j) {
      vF[j] = vS[j] = vP[j] = 0.f;
      if (j < a.nL && tid < a.L[j].nC) {
        const RecLayer& L = a.L[j]; const int nC = L.nC, NO = 2 * nC;
        vF[j] = L.X[r * NO + tid]; vS[j] = L.X[r * NO + nC + tid];
        if (k > 0) vP[j] = L.Y[(r - 1) * NO + tid];
      }
    }
    ldsBarrier();
#pragma unroll
    for (int j = HL_MAX_HIDDEN - 1; j >= 0; --j) if (j < a.nL) {
      const RecLayer& L = LL[j];
      const int nIn = L.nIn, nC = L.nC, NO = 2 * nC;
      const int ldw = LDSW ? NO + 1 : NO, wOff = LDSW ? recLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * ldw + o]; else return gWj[(size_t)i * ldw + o]; };
      float dLdO = 0.f;
      if (tid < nC) {
        const float eTop = sTop[cur][tid];
        if (L.hasRes) { L.Rd[r * L.ldR + tid] = eTop; sRes[tid] = tid < L.resW ? eTop * wr[j] : 0.f; }
        dLdO = eTop + (k < T ? sRec[j][tid] : 0.f);
        sDS[tid] = dLdO * vF[j] * (1.f - vS[j] * vS[j]);                         // 1) dLdS
      }
      ldsBarrier();
      float fp = 0.f;
      if (tid < nC && k > 0) {                                                   // 2) dLdFprevOut = Wsr dLdS
#pragma unroll 8
        for (int o = 0; o < nC; ++o) fp += wAt(nIn + tid, nC + o) * sDS[o];
      }
      if (tid < nC) {
        sFP[tid] = fp;
        sDF[tid] = ((vS[j] - vP[j]) * dLdO + fp * vP[j]) * vF[j] * (1.f - vF[j]);   // 3) dLdF
        L.D[r * NO + tid] = sDF[tid]; L.D[r * NO + nC + tid] = sDS[tid];
      }
      ldsBarrier();
      // backprop to the block input: Wff dLdF + Wsf dLdS (+ the residual path); not below the first layer
      if (j > 0) for (int i = tid; i < nIn; i += 256) {
        float e1 = 0.f, e2 = 0.f;
#pragma unroll 8
        for (int o = 0; o < nC; ++o) { e1 += wAt(i, o) * sDF[o]; e2 += wAt(i, nC + o) * sDS[o]; }
        sTop[cur ^ 1][i] = ((L.hasRes && i < L.resW ? sRes[i] : 0.f) + e1) + e2;
      }
      float rec = 0.f;
      if (k > 0 && tid < nC) {                                                   // 4) dLdprevOut
        rec = (1.f - vF[j]) * dLdO + vF[j] * sFP[tid];
        float g = 0.f;
#pragma unroll 8
        for (int o = 0; o < nC; ++o) g += wAt(nIn + tid, o) * sDF[o];
        rec += g;
      }
      ldsBarrier();
      if (tid < nC) sRec[j][tid] = rec;
      cur ^= 1;
    }
    ldsBarrier();
  }
}

static size_t recLdsBytes(const RecArgs& a) {
  size_t fl = 0;
  for (int j = 0; j < a.nL; ++j) fl += (size_t)(a.L[j].nIn + a.L[j].nC) * (a.gates * a.L[j].nC + 1);
  return fl * sizeof(float);
}
template <class K> static hipError_t recLaunch(K kernel, const RecArgs& a, size_t lds, size_t* attr, hipStream_t s) {
  if (lds > *attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    *attr = lds;
  }
  hipLaunchKernelGGL(kernel, dim3(a.B), dim3(256), lds, s, a);
  return hipGetLastError();
}
// (static LDS of the kernels comes on top of the weights; 160 KB per workgroup)
hipError_t launch_rec_forward(const RecArgs& a, hipStream_t s) {
  static size_t attr[4] = {0, 0, 0, 0}; const size_t lds = recLdsBytes(a); const bool fit = lds <= 120 * 1024;
  if (a.gates == 2) return fit ? recLaunch(mgu_forward_kernel<true>, a, lds, &attr[0], s) : recLaunch(mgu_forward_kernel<false>, a, 0, &attr[1], s);
  return fit ? recLaunch(rec_forward_kernel<true>, a, lds, &attr[2], s) : recLaunch(rec_forward_kernel<false>, a, 0, &attr[3], s);
}
hipError_t launch_rec_backward(const RecArgs& a, hipStream_t s) {
  static size_t attr[4] = {0, 0, 0, 0}; const size_t lds = recLdsBytes(a); const bool fit = lds <= 120 * 1024;
  if (a.gates == 2) return fit ? recLaunch(mgu_backward_kernel<true>, a, lds, &attr[0], s) : recLaunch(mgu_backward_kernel<false>, a, 0, &attr[1], s);
  return fit ? recLaunch(rec_backward_kernel<true>, a, lds, &attr[2], s) : recLaunch(rec_backward_kernel<false>, a, 0, &attr[3], s);
}

}  // namespace hl
