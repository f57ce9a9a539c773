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
#include "head_rows.h"      // (tail_dev.h; the head of the one-launch step: lstm32_step_wave_kernel)
#include "rec_dev.h"

namespace hl {

#define REC_MAXC 64       // cells per layer (4 gates x 64 = 256 threads)
#define REC_MAXIN 256     // inputs of the first layer
#define REC_STATES 4608   // window states kept in LDS (e.g. 18 steps x 256 state components)
// the one-gate-per-thread kernels (rec_*, mgu_*, rnn_*: every shape the specialised ones do not take) loop over gates and cells:
#define REC_GENC 256      // cells per layer there
#define REC_GENIN 1024    // inputs of the first layer there (stacked observations, or the output of a convolutional stack)

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the global stores of the step
// (the rows kept for the backward pass / the dW launch), ~1 us each, and nothing in these kernels reads them back
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// End of a kernel prologue: every global load issued so far has landed.  Values fetched once before the step loops (biases,
// residual parameters, state scales) otherwise look "possibly in flight" at the loop head, and the compiler guards each use
// inside the loop with s_waitcnt vmcnt(0) -- which also waits for every row store of the previous layer-step to be acknowledged.
__device__ __forceinline__ void vmDrain() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt untouched


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

#ifdef REC_STAMPS
__device__ unsigned long long recStamps[256];
#define RSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 256) recStamps[i] = wall_clock64(); } while (0)
// (the one-launch step: first sample's workgroup -- block 1 behind a rider --, wavefront 0 / wavefront 1)
#define SSTAMP(i) do { if (b == 0 && threadIdx.x == 0 && (i) < 256) recStamps[i] = wall_clock64(); } while (0)
#define SSTAMP1(i) do { if (b == 0 && threadIdx.x == 64 && (i) < 256) recStamps[i] = wall_clock64(); } while (0)
#else
#define RSTAMP(i) do {} while (0)
#define SSTAMP(i) do {} while (0)
#define SSTAMP1(i) do {} while (0)
#endif
template <bool LDSW>
__global__ __launch_bounds__(256) void rec_forward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sBuf[2][REC_GENIN];                    // input of the current layer / output of the current block
  __shared__ float sPrevOut[HL_MAX_HIDDEN][REC_GENC], sPrevSt[HL_MAX_HIDDEN][REC_GENC];
  __shared__ float sX[4 * REC_GENC];
  const int b = blockIdx.x, tid = threadIdx.x;
  RSTAMP(0);
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
  RSTAMP(1);
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  RSTAMP(2);
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
  const int dIn = a.L[0].nIn;
  const bool plain = a.Xin == nullptr && a.nApp == 0;      // the input of a step is that step's observed state
  const bool preload = plain && nSteps * a.dS <= REC_STATES;
  if (preload) for (int e = tid; e < nSteps * a.dS; e += 256) {
    const int kk = e / a.dS, i = e - kk * a.dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * a.dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  vmDrain(); ldsBarrier();
  RSTAMP(3);
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    const long long sl = slot - T + k;
    if (!plain) { for (int e = tid; e < dIn; e += 256) sBuf[0][e] = recInputAt(a, acting, b, slot, t, T, nextRow, k, e); }
    else if (tid < a.dS) {
      if (preload) sBuf[0][tid] = sStates[k * a.dS + tid];
      else { const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)sl * a.dS + tid]; sBuf[0][tid] = (raw - sMean) * sScale; }
    }
    ldsBarrier();
    RSTAMP(4 + k * 5);
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
      for (int o = tid; o < NO; o += 256) {                // (one gate per thread up to 64 cells, four at 256)
        float acc = o == tid ? bias[j] : W[L.indB + o];
        // (unrolled: the LDS reads of eight terms are in flight together; a rolled loop pays the LDS latency per term)
#pragma unroll 8
        for (int i = 0; i < nIn; ++i) acc += in[i] * wAt(i, o);
        if (k > 0) {
#pragma unroll 8
          for (int i = 0; i < nC; ++i) acc += sPrevOut[j][i] * wAt(nIn + i, o);
        }
        if (o >= nC) acc = recSigm(acc);                   // the gates overwrite their inputs
        sX[o] = acc;
        if (store) L.X[r * NO + o] = acc;
      }
      ldsBarrier();
      if (j < 2) RSTAMP(4 + k * 5 + 1 + 2 * j);
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
      if (j < 2) RSTAMP(4 + k * 5 + 2 + 2 * j);
      if (tid < nC) { sPrevOut[j][tid] = out; sPrevSt[j][tid] = st; }
      cur ^= 1;
    }
    const int nCl = a.L[a.nL - 1].nC;
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    ldsBarrier();
  }
  RSTAMP(250);
}

// ---- LSTM forward, weights resident in LDS: the version the step uses whenever they fit --------------------------------
// Measured on the kernel above (device time stamps, 2 x 32 cells): a layer-step cost 1.1-1.3 us for the gate sums -- one
// thread per gate walking 36 / 64 terms, eight LDS reads in flight at a time, 0.11 us per batch of eight -- 0.4 us for the
// cell update behind a barrier of its own, and 0.5 us of per-step barriers.  Here
//   * the weights sit TRANSPOSED in LDS, one row per gate: [W_in column (padded to 4) | W_rec column (padded to 4) | bias,0,0,0]
//     with a row pitch whose quarter is odd, so that ds_read_b128 of 16 neighbouring gates touch all 64 banks once;
//   * the operand is ONE vector per layer, [input | previous output | 1,0,0,0], double-buffered over the steps (this step's
//     output goes into the other copy), which is also exactly the row stored for the weight-gradient contraction;
//   * P = 8 / 4 / 2 / 1 lanes share a gate (<= 8 / 16 / 32 / 64 cells), each lane reads four 16-byte chunks per round with all
//     eight reads in flight, partial sums joined by shuffles;
//   * the four gates of a cell sit in neighbouring lanes, so the cell update gathers them by shuffles instead of LDS + barrier:
//     ONE barrier per layer-step, none per step (the next step's state is written into the idle copy during the step).
// Sums are formed in a different order than in the oracle (four partial sums per lane, bias last): 1e-7-level differences.
struct LstmGeo { int inPad, recPad, nT, ld; };
__host__ __device__ __forceinline__ LstmGeo lstmGeo(int nIn, int nC) {
  LstmGeo g; g.inPad = (nIn + 3) & ~3; g.recPad = (nC + 3) & ~3; g.nT = g.inPad + g.recPad + 4;
  g.ld = ((g.nT >> 2) & 1) ? g.nT : g.nT + 4;
  return g;
}
#define LSTM_VEC (REC_MAXIN + REC_MAXC + 8)
// NL / NC: number of layers / cells of every layer known at compile time (0: read from the arguments) -- with the general
// eight-layer body the loop invariants alone are 450 spilled scalars and each layer-step some 600 instructions
template <int NL, int NC>
__global__ __launch_bounds__(256) void lstm_forward_lds_kernel(RecArgs a) {
  constexpr int MAXL = NL ? NL : HL_MAX_HIDDEN;
  const int nL = NL ? NL : a.nL;
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ __attribute__((aligned(16))) float sA[HL_MAX_HIDDEN][2][LSTM_VEC];
  __shared__ float sStates[REC_STATES];
  const int b = blockIdx.x, tid = threadIdx.x;
  RSTAMP(0);
  const bool acting = a.actStates != nullptr;
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  RecLayer LL[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) LL[j] = a.L[j];
  RSTAMP(1);
  // weights, transposed (coalesced 16-byte global reads along the gates of one input row, scattered LDS writes)
  {
    int off = 0;
    for (int j = 0; j < nL; ++j) {
      const RecLayer& L = a.L[j];
      const int nIn = L.nIn, nC = NC ? NC : L.nC, NO = 4 * nC, rows = nIn + nC, q4 = nC;          // q4: float4 per row (NO / 4)
      const LstmGeo g = lstmGeo(nIn, nC);
      const float4* src = reinterpret_cast<const float4*>(W + L.indW);                  // (indW is a multiple of 4: checked by the launcher)
      const int total = rows * q4;
      for (int e0 = tid; e0 < total; e0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; v[u] = src[e < total ? e : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + 256 * u;
          if (e < total) {
            const int i = e / q4, o = 4 * (e - i * q4), ti = i < nIn ? i : g.inPad + (i - nIn);
            float* d = sW + off + o * g.ld + ti;
            d[0] = v[u].x; d[g.ld] = v[u].y; d[2 * g.ld] = v[u].z; d[3 * g.ld] = v[u].w;
          }
        }
      }
      for (int o = tid; o < NO; o += 256) {
        float* d = sW + off + o * g.ld;
        for (int i = nIn; i < g.inPad; ++i) d[i] = 0.f;
        for (int i = g.inPad + nC; i < g.inPad + g.recPad; ++i) d[i] = 0.f;
        d[g.nT - 4] = W[L.indB + o]; d[g.nT - 3] = 0.f; d[g.nT - 2] = 0.f; d[g.nT - 1] = 0.f;
      }
      for (int i = tid; i < 2 * LSTM_VEC; i += 256) {                                  // operand vectors: zero, then the constant 1
        const int c = i / LSTM_VEC, e = i - c * LSTM_VEC;
        sA[j][c][e] = e == g.nT - 4 ? 1.f : 0.f;
      }
      off += NO * g.ld;
    }
  }
  RSTAMP(2);
  // this thread's role per layer: cell, gate, part; the parameters of its cell
  float wr[MAXL], br[MAXL], prevSt[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) {
    wr[j] = 0.f; br[j] = 0.f; prevSt[j] = 0.f;
    if (j < nL) {
      const RecLayer& L = a.L[j];
      const int nCj = NC ? NC : L.nC;
      const int P = nCj <= 8 ? 8 : (nCj <= 16 ? 4 : (nCj <= 32 ? 2 : 1)), c = tid / (4 * P);
      if (L.hasRes && c < L.resW && c < nCj) { wr[j] = W[L.indWr + c]; br[j] = W[L.indBr + c]; }
    }
  }
  const bool preload = nSteps * a.dS <= REC_STATES;
  if (preload) for (int e = tid; e < nSteps * a.dS; e += 256) {
    const int kk = e / a.dS, i = e - kk * a.dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * a.dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  auto stateOf = [&](int k) -> float {     // standardised state component `tid` of step k (Episode::standardizedState, Episode.h:172-183)
    if (preload) return sStates[k * a.dS + tid];
    const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)(slot - T + k) * a.dS + tid];
    return (raw - sMean) * sScale;
  };
  vmDrain(); ldsBarrier();
  if (tid < a.dS) sA[0][0][tid] = stateOf(0);
  ldsBarrier();
  RSTAMP(3);
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    const int cb = k & 1;
    if (k + 1 < nSteps && tid < a.dS) sA[0][cb ^ 1][tid] = stateOf(k + 1);             // (that copy was last read a step ago)
    RSTAMP(4 + k * 5);
    int off = 0;
#pragma unroll
    for (int j = 0; j < MAXL; ++j) if (j < nL) {
      const RecLayer& L = LL[j];
      const int nIn = (NC && j > 0) ? NC : L.nIn, nC = NC ? NC : L.nC, NO = 4 * nC;
      const LstmGeo g = lstmGeo(nIn, nC);
      const int P = nC <= 8 ? 8 : (nC <= 16 ? 4 : (nC <= 32 ? 2 : 1)), G = 4 * P;
      const int c = tid / G, gate = (tid & (G - 1)) / P, part = tid & (P - 1), o = gate * nC + c;
      const float* vec = sA[j][cb];
      if (store) {
        if (tid < nIn + nC) L.A[r * L.ldA + tid] = vec[tid < nIn ? tid : g.inPad + (tid - nIn)];   // (nIn + nC <= 256 + 64: second round below)
        if (tid + 256 < nIn + nC) L.A[r * L.ldA + tid + 256] = vec[tid + 256 < nIn ? tid + 256 : g.inPad + (tid + 256 - nIn)];
      }
      float acc = 0.f;
      if (c < nC) {
        const float4* row4 = reinterpret_cast<const float4*>(sW + off + o * g.ld);
        const float4* vec4 = reinterpret_cast<const float4*>(vec);
        // the bias chunk [b,0,0,0] x [1,0,0,0] is taken as one scalar (lane part 0), so that 64 terms are exactly eight chunks
        // per lane at two lanes per gate: ONE round of sixteen 16-byte reads in flight
        const int nCh = (g.nT >> 2) - 1;
        float p0 = part == 0 ? (sW + off + o * g.ld)[g.nT - 4] : 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        for (int c0 = part; c0 < nCh; c0 += 8 * P) {
          float4 w[8], x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ch = c0 + u * P, cc = ch < nCh ? ch : part;
            w[u] = row4[cc]; x[u] = vec4[cc];
            if (ch >= nCh) w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { p0 += w[u].x * x[u].x; p1 += w[u].y * x[u].y; p2 += w[u].z * x[u].z; p3 += w[u].w * x[u].w; }
        }
        acc = (p0 + p1) + (p2 + p3);
      }
      if (P > 1) acc += __shfl_xor(acc, 1, 64);
      if (P > 2) acc += __shfl_xor(acc, 2, 64);
      if (P > 4) acc += __shfl_xor(acc, 4, 64);
      if (gate > 0) acc = recSigm(acc);                    // the gates overwrite their inputs
      if (store && part == 0 && c < nC) L.X[r * NO + o] = acc;
      const int base = (tid & 63) & ~(G - 1);
      const float x0 = __shfl(acc, base, 64), x1 = __shfl(acc, base + P, 64), x2 = __shfl(acc, base + 2 * P, 64), x3 = __shfl(acc, base + 3 * P, 64);
      if ((tid & (G - 1)) == 0 && c < nC) {
        const float st = x0 * x1 + prevSt[j] * x2;          // (prevSt is 0 at the first step of the window)
        const float co = actEval(HL_FUNC_TANH, st);
        const float out = x3 * co;
        prevSt[j] = st;
        if (store) { L.Y[r * NO + c] = out; L.Y[r * NO + nC + c] = st; L.Y[r * NO + 2 * nC + c] = co; }
        float blk = out;                                   // ParametricResidualLayer::forward (Layers.h:347-361)
        if (L.hasRes && c < L.resW) blk += vec[c] * wr[j] + br[j];
        sA[j][cb ^ 1][g.inPad + c] = out;                  // recurrent input of the next step
        if (j + 1 < nL) sA[j + 1][cb][c] = blk;
        else {
          if (k == T) a.Yout[(size_t)b * a.ldY + c] = blk;
          if (k == T + 1) a.Yout[(size_t)nextRow * a.ldY + c] = blk;
        }
      }
      ldsBarrier();
      if (j < 2) RSTAMP(4 + k * 5 + 2 + 2 * j);
      off += NO * g.ld;
    }
  }
  RSTAMP(250);
}

template <bool LDSW>
__global__ __launch_bounds__(256) void rec_backward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_GENIN];                    // error w.r.t. the output of the current block (from above, same step)
  __shared__ float sRec[HL_MAX_HIDDEN][REC_GENC];          // error w.r.t. this step's LSTM output coming from step k+1
  __shared__ float sNxtSt[HL_MAX_HIDDEN][REC_GENC], sNxtF[HL_MAX_HIDDEN][REC_GENC];
  __shared__ float sD[4 * REC_GENC], sRes[REC_GENC], sRecNew[REC_GENC];
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
  vmDrain(); ldsBarrier();
  // rows of the steps this sample does not have: zero deltas (their stale inputs then add nothing to the gradients)
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      for (int o = tid; o < 4 * L.nC; o += 256) L.D[r * 4 * L.nC + o] = 0.f;
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
// ---- LSTM back-propagation through the window, weights AND the window's stored activations resident in LDS ----------------
// Same findings as for the forward kernel; in addition the per-step reads of the stored gates sat behind the acknowledgement
// of the previous step's delta stores (one in-order memory counter).  Here the gates and states of the whole window are
// fetched once into LDS, so the step loop only stores; [W_in; W_rec] rows have a pitch of 16 (mod 64) floats so that the
// 16-byte reads of four rows x four lanes touch every bank once; two barriers per layer-step instead of three.
__host__ __device__ __forceinline__ int lstmBwdPitch(int NO) { return NO + ((16 - NO % 64) + 64) % 64; }
template <int NL>
__global__ __launch_bounds__(256) void lstm_backward_lds_kernel(RecArgs a) {
  constexpr int MAXL = NL ? NL : HL_MAX_HIDDEN;
  const int nL = NL ? NL : a.nL;
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_MAXIN];                    // error w.r.t. the output of the current block (from above, same step)
  __shared__ float sRec[MAXL][REC_MAXC];                   // error w.r.t. this step's LSTM output coming from step k+1
  __shared__ float sNxtSt[MAXL][REC_MAXC], sNxtF[MAXL][REC_MAXC];
  __shared__ __attribute__((aligned(16))) float sD[4 * REC_MAXC];
  __shared__ float sRes[REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  RecLayer LL[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) LL[j] = a.L[j];
  // weights: rows of [W_in; W_rec] as in global memory, padded pitch
  int wOffs[MAXL], aOffs[MAXL];
  int off = 0, actPitch = 0;
#pragma unroll
  for (int j = 0; j < MAXL; ++j) {
    wOffs[j] = off; aOffs[j] = actPitch;
    if (j < nL) {
      const RecLayer& L = LL[j];
      const int nC = L.nC, NO = 4 * nC, rows = L.nIn + nC, ldb = lstmBwdPitch(NO), total = rows * nC;
      const float4* src = reinterpret_cast<const float4*>(W + L.indW);
      for (int e0 = tid; e0 < total; e0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; v[u] = src[e < total ? e : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + 256 * u;
          if (e < total) { const int i = e / nC, q = e - i * nC; *reinterpret_cast<float4*>(sW + off + i * ldb + 4 * q) = v[u]; }
        }
      }
      off += rows * ldb; actPitch += 6 * nC;
    }
  }
  // stored activations of the window, per (step, layer): [cell input | input, forget, output gate | state | tanh(state)]
  float* sAct = sW + off;
  {
    const int total = (T + 1) * actPitch;
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u, ee = e < total ? e : 0, k = ee / actPitch, q = ee - k * actPitch;
        const long long r = (long long)b * a.K + k;
        const float* p = nullptr;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) if (j < nL && q >= aOffs[j] && q < aOffs[j] + 6 * LL[j].nC) {
          const int x = q - aOffs[j], nC = LL[j].nC;
          p = x < 4 * nC ? LL[j].X + r * 4 * nC + x : LL[j].Y + r * 4 * nC + nC + (x - 4 * nC);
        }
        v[u] = *p;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < total) sAct[e] = v[u]; }
    }
  }
  float wr[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) { wr[j] = 0.f; if (j < nL && LL[j].hasRes && tid < LL[j].resW) wr[j] = W[LL[j].indWr + tid]; }
  const int nCl = a.L[nL - 1].nC;      // (from the arguments: a run-time index into LL would put the whole array on the stack -- 976 bytes of scratch per lane in the any-depth variant until round 5)
  const float dres = tid < nCl ? a.Dres[(size_t)b * a.ldD + tid] : 0.f;
  vmDrain(); ldsBarrier();
  // rows of the steps this sample does not have: zero deltas (their stale inputs then add nothing to the gradients)
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < nL; ++j) {
      const RecLayer& L = a.L[j];
      if (tid < 4 * L.nC) L.D[r * 4 * L.nC + tid] = 0.f;
      if (L.hasRes && tid < L.nC) L.Rd[r * L.ldR + tid] = 0.f;
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    if (tid < nCl) sTop[0][tid] = k == T ? dres : 0.f;
    ldsBarrier();
#pragma unroll
    for (int j = MAXL - 1; j >= 0; --j) if (j < nL) {
      const RecLayer& L = LL[j];
      const int nIn = L.nIn, nC = L.nC, NO = 4 * nC, ldb = lstmBwdPitch(NO);
      if (tid < nC) {
        const float* act = sAct + k * actPitch + aOffs[j];
        const float eTop = sTop[cur][tid];
        // ParametricResidualLayer::backward (Layers.h:363-393): the delta passes to the LSTM output, and through w to the block input
        if (L.hasRes) { L.Rd[r * L.ldR + tid] = eTop; sRes[tid] = tid < L.resW ? eTop * wr[j] : 0.f; }
        const float D = eTop + (k < T ? sRec[j][tid] : 0.f);
        // LSTMLayer::backward (Layer_LSTM.h:127-165)
        const float cellInpt = act[tid], IG = act[nC + tid], FG = act[2 * nC + tid], OG = act[3 * nC + tid], co = act[5 * nC + tid];
        const float prevSt = k > 0 ? (act - actPitch)[4 * nC + tid] : 0.f;
        const float diff = (1.f - co * co) * D;
        const float sd = diff * OG + (k < T ? sNxtSt[j][tid] * sNxtF[j][tid] : 0.f);
        const float d0 = IG * sd;
        const float d1 = IG * (1.f - IG) * cellInpt * sd;
        const float d2 = k > 0 ? FG * (1.f - FG) * prevSt * sd : 0.f;
        const float d3 = OG * (1.f - OG) * D * co;
        sD[tid] = d0; sD[nC + tid] = d1; sD[2 * nC + tid] = d2; sD[3 * nC + tid] = d3;
        L.D[r * NO + tid] = d0; L.D[r * NO + nC + tid] = d1; L.D[r * NO + 2 * nC + tid] = d2; L.D[r * NO + 3 * nC + tid] = d3;
        sNxtSt[j][tid] = sd; sNxtF[j][tid] = FG;
      }
      ldsBarrier();
      // Layer::backward (Layers.h:123-188): one row of [W_in; W_rec] per group of four lanes, 16-byte chunks of the row and of
      // the deltas, eight of each in flight per lane; rows 0..nIn-1 give the error of the block below (skipped under the first
      // layer), rows nIn.. the error handed to the previous step
      {
        const int part = tid & 3, row0 = j > 0 ? 0 : nIn, nRow = nIn + (k > 0 ? nC : 0), nCh = nC;
        const float4* d4 = reinterpret_cast<const float4*>(sD);
        for (int i0 = row0; i0 < nRow; i0 += 64) {
          const int i = i0 + (tid >> 2);
          float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
          if (i < nRow) {
            const float4* row4 = reinterpret_cast<const float4*>(sW + wOffs[j] + i * ldb);
            for (int u0 = 0; u0 < nCh; u0 += 32) {
              float4 w[8], d[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int ch = u0 + part + 4 * u, cc = ch < nCh ? ch : part;
                w[u] = row4[cc]; d[u] = d4[cc];
                if (ch >= nCh) w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) { p0 += w[u].x * d[u].x; p1 += w[u].y * d[u].y; p2 += w[u].z * d[u].z; p3 += w[u].w * d[u].w; }
            }
          }
          float e = (p0 + p1) + (p2 + p3);
          e += __shfl_xor(e, 1, 64); e += __shfl_xor(e, 2, 64);
          if (part == 0 && i < nRow) {
            if (i < nIn) sTop[cur ^ 1][i] = (L.hasRes && i < L.resW ? sRes[i] : 0.f) + e;
            else sRec[j][i - nIn] = e;                       // (read above, before the barrier; not used at k == 0)
          }
        }
      }
      ldsBarrier();
      cur ^= 1;
    }
  }
}

template <bool LDSW, int NL>
__global__ __launch_bounds__(256) void mgu_forward_kernel(RecArgs a) {
  constexpr int MAXL = NL ? NL : HL_MAX_HIDDEN;
  const int nL = NL ? NL : a.nL;
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sBuf[2][REC_GENIN];
  __shared__ float sPrevOut[MAXL][REC_GENC];
  __shared__ float sF[REC_GENC], sS[REC_GENC];
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
  RecLayer LL[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float bias[MAXL], wr[MAXL], br[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) {
    bias[j] = 0.f; wr[j] = 0.f; br[j] = 0.f;
    if (j < nL) {
      const RecLayer& L = a.L[j];
      if (tid < 2 * L.nC) bias[j] = W[L.indB + tid];
      if (L.hasRes && tid < L.resW) { wr[j] = W[L.indWr + tid]; br[j] = W[L.indBr + tid]; }
    }
  }
  const int dIn = a.L[0].nIn;
  const bool plain = a.Xin == nullptr && a.nApp == 0;      // the input of a step is that step's observed state
  const bool preload = plain && nSteps * a.dS <= REC_STATES;
  if (preload) for (int e = tid; e < nSteps * a.dS; e += 256) {
    const int kk = e / a.dS, i = e - kk * a.dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * a.dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  vmDrain(); ldsBarrier();
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    if (!plain) { for (int e = tid; e < dIn; e += 256) sBuf[0][e] = recInputAt(a, acting, b, slot, t, T, nextRow, k, e); }
    else if (tid < a.dS) {
      if (preload) sBuf[0][tid] = sStates[k * a.dS + tid];
      else { const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)(slot - T + k) * a.dS + tid]; sBuf[0][tid] = (raw - sMean) * sScale; }
    }
    ldsBarrier();
    int cur = 0;
#pragma unroll
    for (int j = 0; j < MAXL; ++j) if (j < nL) {
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
      // (one gate per thread up to 128 cells, two at 256; the state gates' input sums wait in sS for the forget gates)
      for (int o = tid; o < NO; o += 256) {
        float acc = o == tid ? bias[j] : W[L.indB + o];
#pragma unroll 8
        for (int i = 0; i < nIn; ++i) acc += in[i] * wAt(i, o);
        if (o < nC) {          // forget gate
          if (k > 0) {
#pragma unroll 8
            for (int i = 0; i < nC; ++i) acc += wAt(nIn + i, o) * sPrevOut[j][i];
          }
          acc = recSigm(acc);
          sF[o] = acc;
          if (store) L.X[r * NO + o] = acc;
        } else sS[o - nC] = acc;
      }
      ldsBarrier();
      for (int o = nC + tid; o < NO; o += 256) {   // cell state
        float acc = sS[o - nC];
        if (k > 0) {
#pragma unroll 8
          for (int i = 0; i < nC; ++i) acc += wAt(nIn + i, o) * sPrevOut[j][i] * sF[i];
        }
        acc = actEval(HL_FUNC_TANH, acc);
        sS[o - nC] = acc;
        if (store) L.X[r * NO + o] = acc;
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
    const int nCl = a.L[nL - 1].nC;
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    ldsBarrier();
  }
}

template <bool LDSW, int NL>
__global__ __launch_bounds__(256) void mgu_backward_kernel(RecArgs a) {
  constexpr int MAXL = NL ? NL : HL_MAX_HIDDEN;
  const int nL = NL ? NL : a.nL;
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_GENIN];
  __shared__ float sRec[MAXL][REC_GENC];          // dLdprevOut handed from step k+1 to step k
  __shared__ float sDF[REC_GENC], sDS[REC_GENC], sFP[REC_GENC], sRes[REC_GENC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  // layer descriptors copied out of the kernel-argument segment once (inside the step loops every field access was a scalar
  // load of its own)
  RecLayer LL[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) LL[j] = a.L[j];
  if constexpr (LDSW) recStageWeights(a, sW, tid);
  float wr[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; ++j) { wr[j] = 0.f; if (j < nL && a.L[j].hasRes && tid < a.L[j].resW) wr[j] = W[a.L[j].indWr + tid]; }
  vmDrain(); ldsBarrier();
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < nL; ++j) {
      const RecLayer& L = a.L[j];
      for (int o = tid; o < 2 * L.nC; o += 256) L.D[r * 2 * L.nC + o] = 0.f;
      if (L.hasRes && tid < L.nC) L.Rd[r * L.ldR + tid] = 0.f;
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    const int nCl = a.L[nL - 1].nC;
    if (tid < nCl) sTop[0][tid] = k == T ? a.Dres[(size_t)b * a.ldD + tid] : 0.f;
    float vF[MAXL], vS[MAXL], vP[MAXL];
#pragma unroll
    for (int j = 0; j < MAXL; ++j) {
      vF[j] = vS[j] = vP[j] = 0.f;
      if (j < nL && tid < a.L[j].nC) {
        const RecLayer& L = a.L[j]; const int nC = L.nC, NO = 2 * nC;
        vF[j] = L.X[r * NO + tid]; vS[j] = L.X[r * NO + nC + tid];
        if (k > 0) vP[j] = L.Y[(r - 1) * NO + tid];
      }
    }
    ldsBarrier();
#pragma unroll
    for (int j = MAXL - 1; j >= 0; --j) if (j < nL) {
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

// ---- LSTM, two layers of 32 cells (settings/RACER_RNN.json): ONE WAVEFRONT PER (SAMPLE, LAYER), weights in registers ----------
// The kernels above spend a layer-step (128 gates x 64 terms = 8 k multiply-adds) mostly on LDS round trips for weights and
// operand, cross-lane joins and workgroup barriers: ~1.1 us, 38 + 41 us per window forward + BPTT.  Here a workgroup of two
// wavefronts owns a sample, one wavefront per layer:
//   forward   lane l of a layer's wavefront holds the two gate columns l and l + 64 of [W_in; W_rec] in registers; the operand
//             [input | previous output] lives in LDS and is read as broadcast 16-byte chunks; the two dot products of a lane
//             run as packed fp32 FMAs on four independent accumulators, no cross-lane join; the four gates of a cell meet
//             through one 32-lane shuffle.  Layer 0 never depends on layer 1, so its wavefront runs one step AHEAD: at
//             iteration i the first works on step i, the second on step i - 1 (software pipeline, one barrier per iteration).
//   backward  lane i holds ROW i of [W_in; W_rec] (128 values), the 128 gate deltas of the layer-step are broadcast from LDS.
//             Here the TOP layer never depends on the one below, so its wavefront runs one step ahead (k = T - i).
// Sigmoid / tanh use the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32 + one Newton step: ~1e-7 relative, inside
// the tolerance these layers are tested to); the kernels above keep libm's.  Rows written for the weight-gradient launch
// are the same as those of the kernels above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// lanes 0..31 receive the value lane + 32 holds: one v_permlane32_swap (gfx950) instead of a trip through the LDS crossbar
// (ds_bpermute).  What lanes 32..63 receive is not used by any caller.
__device__ __forceinline__ float fromUpperHalf(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[1]);
}
__device__ __forceinline__ void waveLdsSync() { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void pairBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float fastRcp(float d) { float r = __builtin_amdgcn_rcpf(d); return fmaf(fmaf(-d, r, 1.0f), r, r); }
__device__ __forceinline__ float fastSigm(float in) {      // Sigm::_eval with safeExp cut at 8, one exponential for both branches
  const float ex = __expf(fmaxf(-8.f, -fabsf(in))), r = fastRcp(1.f + ex);
  return in > 0.f ? r : ex * r;
}
__device__ __forceinline__ float fastTanh(float in) {      // Tanh::_eval (Functions.h:104-113)
  const float e = __expf(-2.f * fabsf(in)), y = (1.f - e) * fastRcp(1.f + e);
  return in > 0.f ? y : -y;
}

// one layer-step: the two gates of this lane (NT terms each, operand broadcast from LDS), cell update on lanes 0..31, the
// rows kept for the backward pass / the weight gradients
// (LDSACT: gates, state and tanh(state) of the step go to `actRow` in LDS -- the one-launch step keeps them for its own backward pass)
template <int NT, bool LDSACT = false>
__device__ __forceinline__ void lstm32LayerStep(const RecLayer& L, const f32x2 (&w)[NT], f32x2 bias, const float* vec, int inPad, int nInL,
                                                float& prevSt, float wr, float br, float* hNext, float& blkOut, bool store, long long r, int lane,
                                                float* actRow = nullptr) {
  constexpr int NC = 32, NO = 128;
  f32x2 a0 = bias, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
  const f32x4* v4 = reinterpret_cast<const f32x4*>(vec);
  // every operand read before the first FMA: left to itself the compiler reads two 16-byte pieces at a time into the same
  // registers and waits for them (lgkmcnt(0)) in front of each batch of eight FMAs -- seven exposed LDS latencies per layer-step
  f32x4 vq[NT / 4];
#pragma unroll
  for (int q = 0; q < NT / 4; ++q) vq[q] = v4[q];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < NT / 4; ++q) {
    const f32x4 v = vq[q];
    a0 += w[4 * q] * f32x2{v[0], v[0]}; a1 += w[4 * q + 1] * f32x2{v[1], v[1]};
    a2 += w[4 * q + 2] * f32x2{v[2], v[2]}; a3 += w[4 * q + 3] * f32x2{v[3], v[3]};
  }
  const f32x2 acc = (a0 + a1) + (a2 + a3);
  // lanes 0..31: acc = (cell input, forget gate); lanes 32..63: (input gate, output gate)
  const float g0 = lane < 32 ? acc[0] : fastSigm(acc[0]), g1 = fastSigm(acc[1]);
  if (store) { if (LDSACT) { actRow[lane] = g0; actRow[lane + 64] = g1; } else { L.X[r * NO + lane] = g0; L.X[r * NO + lane + 64] = g1; } }
  const float ig = fromUpperHalf(g0), og = fromUpperHalf(g1);
  if (store && lane < nInL + NC)          // the operand row [input | previous output]: A operand of the weight-gradient contraction (one store)
    L.A[r * L.ldA + lane] = vec[lane < nInL ? lane : inPad + (lane - nInL)];
  if (lane < NC) {
    const float st = g0 * ig + prevSt * g1;            // (prevSt is 0 at the first step of the window)
    const float co = fastTanh(st);
    const float out = og * co;
    prevSt = st;
    if (store) { if (LDSACT) { actRow[4 * NC + lane] = st; actRow[5 * NC + lane] = co; } else { L.Y[r * NO + lane] = out; L.Y[r * NO + NC + lane] = st; L.Y[r * NO + 2 * NC + lane] = co; } }
    float blk = out;                                   // ParametricResidualLayer::forward (Layers.h:347-361)
    if (L.hasRes && lane < L.resW) blk += vec[lane] * wr + br;
    hNext[lane] = out;
    blkOut = blk;
  }
}

template <int IN0>     // inputs of the first layer, padded to a multiple of 4 (<= 32)
__global__ __launch_bounds__(128) void lstm32_forward_wave_kernel(RecArgs a) {
  constexpr int NC = 32, NO = 128;
  constexpr int NTMAX = (IN0 + NC) > 2 * NC ? (IN0 + NC) : 2 * NC;    // terms per gate as the unrolled loop walks them (zero weights behind the layer's own)
  __shared__ __attribute__((aligned(16))) float sV0[2][NTMAX];         // layer 0 operand [x_k | h0_{k-1} | 0 ...], double-buffered over the steps
  __shared__ __attribute__((aligned(16))) float sV1[2][2 * NC];        // layer 1 operand [block-0 output of step k | h1_{k-1}]
  __shared__ float sStates[18 * 32];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  RSTAMP(0);
  const int layer = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform
  const bool acting = a.actStates != nullptr;                          // rollout inference: the agent's last states, nothing stored
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  const RecLayer L = a.L[layer];
  const int nIn = a.L[0].nIn, dS = a.dS;
  // this lane's gate columns of its layer: o = lane (cell input | input gate) and lane + 64 (forget | output gate)
  f32x2 w[NTMAX];
  {
    const float* Wl = W + L.indW;
#pragma unroll
    for (int i = 0; i < NTMAX; ++i) {
      int row;
      if (layer == 0) row = i < IN0 ? (i < nIn ? i : -1) : (i < IN0 + NC ? nIn + (i - IN0) : -1);
      else row = i < 2 * NC ? i : -1;
      w[i] = row >= 0 ? f32x2{Wl[(size_t)row * NO + lane], Wl[(size_t)row * NO + lane + 64]} : f32x2{0.f, 0.f};
    }
  }
  const f32x2 bias = {W[L.indB + lane], W[L.indB + lane + 64]};
  const int c = lane & 31;
  float wr = 0.f, br = 0.f;
  if (L.hasRes && c < L.resW) { wr = W[L.indWr + c]; br = W[L.indBr + c]; }
  for (int e = tid; e < nSteps * dS; e += 128) {
    const int kk = e / dS, i = e - kk * dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  for (int i = tid; i < 2 * NTMAX; i += 128) (&sV0[0][0])[i] = 0.f;
  for (int i = tid; i < 4 * NC; i += 128) (&sV1[0][0])[i] = 0.f;
  RSTAMP(1);
  vmDrain(); pairBarrier();
  RSTAMP(2);
  if (layer == 0 && lane < dS) sV0[0][lane] = sStates[lane];
  pairBarrier();
  float prevSt = 0.f;
  for (int it = 0; it <= nSteps; ++it) {
    RSTAMP(4 + it);
    if (layer == 0) {
      const int k = it;
      if (k < nSteps) {
        const int cb = k & 1;
        if (k + 1 < nSteps && lane < dS) sV0[cb ^ 1][lane] = sStates[(k + 1) * dS + lane];       // (that copy was last read a step ago)
        float blk = 0.f;
        lstm32LayerStep<NTMAX>(L, w, bias, sV0[cb], IN0, nIn, prevSt, wr, br, &sV0[cb ^ 1][IN0], blk, !acting && k <= T, (long long)b * a.K + k, lane);
        if (lane < NC) sV1[cb][lane] = blk;
      }
    } else {
      const int k = it - 1;
      if (k >= 0) {
        const int cb = k & 1;
        float blk = 0.f;
        lstm32LayerStep<NTMAX>(L, w, bias, sV1[cb], NC, NC, prevSt, wr, br, &sV1[cb ^ 1][NC], blk, !acting && k <= T, (long long)b * a.K + k, lane);
        if (lane < NC) {
          if (k == T) a.Yout[(size_t)b * a.ldY + lane] = blk;
          if (k == T + 1) a.Yout[(size_t)nextRow * a.ldY + lane] = blk;
        }
      }
    }
    pairBarrier();
  }
  RSTAMP(250);
}

template <int IN0>
__global__ __launch_bounds__(128) void lstm32_backward_wave_kernel(RecArgs a) {
  constexpr int NC = 32, NO = 128, ACT = 6 * NC;           // per (step, layer): [cell input | I | F | O | state | tanh(state)]
  __shared__ __attribute__((aligned(16))) float sD[2][NO]; // gate deltas of the layer-step, per layer
  __shared__ float sAct[2][17 * ACT];
  __shared__ float sTop[2][NC];                            // error w.r.t. block 0's output at step k (ring over k & 1), from the top layer
  __shared__ float sRec[2][NC];                            // error w.r.t. this step's LSTM output coming from step k + 1, per layer
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = 1 - wv;                                    // wavefront 0 = the top layer (one step ahead), wavefront 1 = layer 0
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  const RecLayer L = a.L[j];
  const int nIn0 = a.L[0].nIn;
  // this lane's row of [W_in; W_rec]: top layer row `lane` (0..31 input, 32..63 recurrent); layer 0 row nIn + lane (recurrent
  // part only: nothing is propagated to the network input)
  f32x4 wq[NO / 4];
  {
    const f32x4* rw = reinterpret_cast<const f32x4*>(W + L.indW + (size_t)(j == 1 ? lane : nIn0 + (lane & 31)) * NO);
#pragma unroll
    for (int q = 0; q < NO / 4; ++q) wq[q] = rw[q];
  }
  {   // stored activations of this layer over the window
    const int total = (T + 1) * ACT;
    for (int e = lane; e < total; e += 64) {
      const int k = e / ACT, x = e - k * ACT;
      const long long r = (long long)b * a.K + k;
      sAct[j][e] = x < 4 * NC ? L.X[r * NO + x] : L.Y[r * NO + NC + (x - 4 * NC)];
    }
  }
  const float wr = (L.hasRes && lane < L.resW) ? W[L.indWr + lane] : 0.f;
  const float dres = (j == 1 && lane < NC) ? a.Dres[(size_t)b * a.ldD + lane] : 0.f;
  // rows of the steps this sample does not have: zero deltas
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    L.D[r * NO + lane] = 0.f; L.D[r * NO + lane + 64] = 0.f;
    if (L.hasRes && lane < NC) L.Rd[r * L.ldR + lane] = 0.f;
  }
  vmDrain(); pairBarrier();
  float nxtSt = 0.f, nxtF = 0.f;                           // state error / forget gate of step k + 1 (lanes 0..31)
  for (int it = 0; it <= T + 1; ++it) {
    const int k = T - it + (j == 1 ? 0 : 1);               // the top layer is one step ahead
    if (k >= 0 && k <= T) {
      const long long r = (long long)b * a.K + k;
      // gate deltas on lanes 0..31 (LSTMLayer::backward, Layer_LSTM.h:127-165)
      float res = 0.f;
      if (lane < NC) {
        const float eTop = j == 1 ? (k == T ? dres : 0.f) : sTop[k & 1][lane];
        const float* act = sAct[j] + k * ACT;
        if (L.hasRes) { L.Rd[r * L.ldR + lane] = eTop; res = lane < L.resW ? eTop * wr : 0.f; }
        const float D = eTop + (k < T ? sRec[j][lane] : 0.f);
        const float cellInpt = act[lane], IG = act[NC + lane], FG = act[2 * NC + lane], OG = act[3 * NC + lane], co = act[5 * NC + lane];
        const float prevSt = k > 0 ? (act - ACT)[4 * NC + lane] : 0.f;
        const float diff = (1.f - co * co) * D;
        const float sd = diff * OG + (k < T ? nxtSt * nxtF : 0.f);
        const float d0 = IG * sd;
        const float d1 = IG * (1.f - IG) * cellInpt * sd;
        const float d2 = k > 0 ? FG * (1.f - FG) * prevSt * sd : 0.f;
        const float d3 = OG * (1.f - OG) * D * co;
        sD[j][lane] = d0; sD[j][NC + lane] = d1; sD[j][2 * NC + lane] = d2; sD[j][3 * NC + lane] = d3;
        L.D[r * NO + lane] = d0; L.D[r * NO + NC + lane] = d1; L.D[r * NO + 2 * NC + lane] = d2; L.D[r * NO + 3 * NC + lane] = d3;
        nxtSt = sd; nxtF = FG;
      }
      __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier();    // this wavefront's deltas are in LDS
      if (j == 1 || k > 0) {
        // e_i = sum_o W[i][o] delta[o] for this lane's row (Layer::backward, Layers.h:123-188)
        const f32x4* d4 = reinterpret_cast<const f32x4*>(sD[j]);
        f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f};
        // (the reads of half the deltas before the first FMA of that half -- see lstm32LayerStep; all 32 pieces at once would
        //  take the kernel past 256 VGPRs)
#pragma unroll
        for (int h0 = 0; h0 < NO / 4; h0 += NO / 8) {
          f32x4 dq[NO / 8];
#pragma unroll
          for (int q = 0; q < NO / 8; ++q) dq[q] = d4[h0 + q];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NO / 8; q += 2) {
            const f32x4 da = dq[q], db = dq[q + 1];
            p0 += f32x2{wq[h0 + q][0], wq[h0 + q][1]} * f32x2{da[0], da[1]}; p1 += f32x2{wq[h0 + q][2], wq[h0 + q][3]} * f32x2{da[2], da[3]};
            p2 += f32x2{wq[h0 + q + 1][0], wq[h0 + q + 1][1]} * f32x2{db[0], db[1]}; p3 += f32x2{wq[h0 + q + 1][2], wq[h0 + q + 1][3]} * f32x2{db[2], db[3]};
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const f32x2 ps = (p0 + p1) + (p2 + p3);
        const float e = ps[0] + ps[1];
        if (j == 1) {   // rows 0..31: error of block 0's output (+ the residual path of this layer); rows 32..63: handed to step k - 1
          if (lane < NC) sTop[k & 1][lane] = res + e; else sRec[1][lane - NC] = e;
        } else if (lane < NC) sRec[0][lane] = e;
      }
    }
    pairBarrier();
  }
}

// ---- the whole sample in ONE launch: window forward, output layer + RACER head, back-propagation through time ---------------------
// A sample's chain -- forward over its window, the head of its sampled step (and of step t+1 of a truncated episode end), backward
// over the window -- needs nothing from other samples, so the three launches of the recurrent step (lstm32_forward_wave_kernel,
// the head kernel, lstm32_backward_wave_kernel) are one workgroup's work here: no two launch boundaries (2 x 2.8 us), the gates and
// states of the window stay in LDS instead of going through memory (five of the forward pass's seven row stores per layer-step and
// the backward pass's prologue of dependent loads), the head's replay rows are requested in the prologue and its output-independent
// terms are formed while layer 0 runs its first step alone, the backward pass's weight rows arrive while the head runs.
// Wavefront 0 = layer 0, wavefront 1 = layer 1 in the forward pass (one step apart), the head on wavefront 1 (16-lane rows of
// head_rows.h: row 0 the sampled step, row 1 the next-state step), the top layer's backward on wavefront 0 as in the kernel above.
// Same arithmetic as the three kernels except the output layer and delta_y = delta_out W_out^T, which are plain fp32 dot products in
// index order here (MFMA contractions with K split over wavefronts there).  `extra`: a rider workgroup (launch of 256 threads).
constexpr int LS_LD = 38, LS_LO = 49;        // head rows in LDS: up to 32 dense outputs, 48 outputs in all
// LDS of the head phase, shared by the LSTM and the MGU form: the last block's outputs ([0] sampled step, [1] step t + 1), the error
// w.r.t. them, W_out [32][ldWo <= 40], and the rows HeadRow::compute works on
struct StepHeadLds {
  float (*sYo)[32]; float* sDres; float* sWo; float (*sXo)[LS_LD]; float (*sDelta)[LS_LD]; float (*sMisc)[8];
  double (*sO)[LS_LO]; double* sActMsg; double* sTq; double* sTr;
};
constexpr int STEP_HEAD_LDS = 2 * 32 * 4 + 32 * 4 + 32 * 40 * 4 + 2 * LS_LD * 4 * 2 + 2 * 8 * 4 + 2 * LS_LO * 8 + 2 * 8 + 2 * 2 * 64 * 8;
static_assert(STEP_HEAD_LDS % 8 == 0, "the double rows are 8-byte aligned");
__device__ __forceinline__ StepHeadLds stepHeadLds(unsigned char* p) {      // p: 8-byte aligned
  StepHeadLds S;
  S.sO = reinterpret_cast<double (*)[LS_LO]>(p); p += 2 * LS_LO * 8;
  S.sActMsg = reinterpret_cast<double*>(p); p += 2 * 8;
  S.sTq = reinterpret_cast<double*>(p); S.sTr = S.sTq + 2 * 64; p += 2 * 2 * 64 * 8;
  S.sYo = reinterpret_cast<float (*)[32]>(p); p += 2 * 32 * 4;
  S.sDres = reinterpret_cast<float*>(p); p += 32 * 4;
  S.sWo = reinterpret_cast<float*>(p); p += 32 * 40 * 4;
  S.sXo = reinterpret_cast<float (*)[LS_LD]>(p); p += 2 * LS_LD * 4;
  S.sDelta = reinterpret_cast<float (*)[LS_LD]>(p); p += 2 * LS_LD * 4;
  S.sMisc = reinterpret_cast<float (*)[8]>(p);
  return S;
}
// what the head lanes (wavefront 1, lanes 0..31: 16-lane row `em` = 0 the sampled step, 1 = step t + 1 of a truncated episode end)
// fetch in the kernel's prologue
struct StepHeadRegs { HeadRow<1> hr; float bpv[1]; float bov0, bov1; double beta, Cmax, Cinv; bool rowValid, isNext, live; int em, en; };
__device__ __forceinline__ void stepHeadLoad(StepHeadRegs& R, const HeadArgs& ha, const DevScalars* sc, int wv, int lane, long long slot, int nextRow) {
  R.em = (lane >> 4) & 1; R.en = lane & 15;
  const bool headLane = wv == 1 && lane < 32;
  R.rowValid = headLane && (R.em == 0 || nextRow >= 0); R.isNext = R.rowValid && R.em == 1; R.live = R.rowValid && !R.isNext;
  R.hr.load(ha, R.rowValid, R.isNext, slot, R.en);
  R.bpv[0] = 0.f; R.bov0 = 0.f; R.bov1 = 0.f;
  if (headLane) {
    if (R.en < ha.nSig) R.bpv[0] = ha.params[ha.indBp + R.en];
    if (R.en < ha.nDense) R.bov0 = ha.params[ha.indBo + R.en];
    if (R.en + 16 < ha.nDense) R.bov1 = ha.params[ha.indBo + R.en + 16];
  }
  R.beta = sc->beta; R.Cmax = sc->Cmax; R.Cinv = sc->Cinv;
}
// output layer + head of the sample's rows and the error w.r.t. the last block's output (S.sDres); all of wavefront 1
__device__ __forceinline__ void stepHeadRun(StepHeadRegs& R, const HeadArgs& ha, const StepHeadLds& S, int lane, int b, long long slot, int nextRow) {
  constexpr int NC = 32;
  const int em = R.em, en = R.en, nDense = ha.nDense, ldWo = ha.ldWo;
  if (lane < 32) {      // O[row][o] = y W_out + b_out, o = en and en + 16 (BaseLayer::forward of the output layer)
    float x0 = 0.f, x1 = 0.f;
    const float* y = S.sYo[em];
    const int o0 = en < ldWo ? en : ldWo - 1, o1 = en + 16 < ldWo ? en + 16 : ldWo - 1;
#pragma unroll 8
    for (int cc = 0; cc < NC; ++cc) { const float yv = y[cc]; x0 = fmaf(yv, S.sWo[cc * ldWo + o0], x0); x1 = fmaf(yv, S.sWo[cc * ldWo + o1], x1); }
    x0 += R.bov0; x1 += R.bov1;
    if (en < nDense) { S.sXo[em][en] = x0; S.sO[em][en] = (double)(ha.outFunc == HL_FUNC_LINEAR ? x0 : actEval(ha.outFunc, x0)); }
    if (en + 16 < nDense) { S.sXo[em][en + 16] = x1; S.sO[em][en + 16] = (double)(ha.outFunc == HL_FUNC_LINEAR ? x1 : actEval(ha.outFunc, x1)); }
    if (en < ha.nSig) S.sO[em][nDense + en] = (double)R.bpv[0];      // ParamLayer, Linear
    for (int o = en; o < LS_LD; o += 16) S.sDelta[em][o] = 0.f;
  }
  waveLdsSync();
  R.hr.compute(ha, S.sO[em], S.sDelta[em], S.sXo[em], S.sMisc[em], S.sTq + em * 64, S.sTr + em * 64, R.rowValid, R.isNext, true,
               b, slot, em ? nextRow : b, en, R.beta, R.Cmax, R.Cinv, S.sActMsg[em]);
  waveLdsSync();
  if (lane < NC) {      // delta_y = delta_out W_out^T: error w.r.t. the last block's output at the sampled step
    float e = 0.f;
    for (int o = 0; o < nDense; ++o) e = fmaf(S.sDelta[0][o], S.sWo[lane * ldWo + o], e);
    S.sDres[lane] = e;
  }
}

template <int IN0>
__global__ __launch_bounds__(256) void lstm32_step_wave_kernel(RecArgs a, HeadArgs ha, unsigned long long boundedMask, ExtraArgs extra) {
  constexpr int NC = 32, NO = 128, ACT = 6 * NC;
  constexpr int NTMAX = (IN0 + NC) > 2 * NC ? (IN0 + NC) : 2 * NC;
  constexpr int O_V0 = 2 * 17 * ACT * 4, O_V1 = O_V0 + 2 * NTMAX * 4, O_ST = O_V1 + 2 * 64 * 4, O_D = O_ST + 18 * 32 * 4, O_TOP = O_D + 2 * NO * 4,
                O_REC = O_TOP + 2 * NC * 4, O_HEAD = O_REC + 2 * NC * 4, TOTAL = O_HEAD + STEP_HEAD_LDS;
  static_assert(O_HEAD % 16 == 0, "alignment of the head rows");
  __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL > (int)TAIL_LDS_BYTES ? TOTAL : (int)TAIL_LDS_BYTES];
  if (extra.role) { if (blockIdx.x == 0) { runExtra(extra, smem); return; } if (threadIdx.x >= 128) return; }
  float* sAct = reinterpret_cast<float*>(smem);                                   // [2][17][ACT]: per (layer, step) [cell input | I | F | O | state | tanh(state)]
  float (*sV0)[NTMAX] = reinterpret_cast<float (*)[NTMAX]>(smem + O_V0);
  float (*sV1)[2 * NC] = reinterpret_cast<float (*)[2 * NC]>(smem + O_V1);
  float* sStates = reinterpret_cast<float*>(smem + O_ST);
  float (*sD)[NO] = reinterpret_cast<float (*)[NO]>(smem + O_D);
  float (*sTop)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_TOP);
  float (*sRec)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_REC);
  const StepHeadLds S = stepHeadLds(smem + O_HEAD);

  const int b = blockIdx.x - (extra.role ? 1 : 0), tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = a.bt.t[b]; const long long slot = a.bt.slot[b];
  const int T = min(a.nBPTT, t);
  const int nextRow = a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  const int nIn = a.L[0].nIn, dS = a.dS;
  SSTAMP(0);
  StepHeadRegs R;
  stepHeadLoad(R, ha, a.sc, wv, lane, slot, nextRow);
  {
    // ================================================ forward over the window ==================================================
    const int layer = wv;
    const RecLayer L = a.L[layer];
    f32x2 w[NTMAX];
    {
      const float* Wl = W + L.indW;
#pragma unroll
      for (int i = 0; i < NTMAX; ++i) {
        int row;
        if (layer == 0) row = i < IN0 ? (i < nIn ? i : -1) : (i < IN0 + NC ? nIn + (i - IN0) : -1);
        else row = i < 2 * NC ? i : -1;
        w[i] = row >= 0 ? f32x2{Wl[(size_t)row * NO + lane], Wl[(size_t)row * NO + lane + 64]} : f32x2{0.f, 0.f};
      }
    }
    const f32x2 bias = {W[L.indB + lane], W[L.indB + lane + 64]};
    const int c = lane & 31;
    float wr = 0.f, br = 0.f;
    if (L.hasRes && c < L.resW) { wr = W[L.indWr + c]; br = W[L.indBr + c]; }
    for (int e = tid; e < nSteps * dS; e += 128) {
      const int kk = e / dS, i = e - kk * dS;
      sStates[e] = (a.rp.S[(size_t)(slot - T + kk) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
    }
    for (int e = tid; e < NC * ha.ldWo; e += 128) S.sWo[e] = ha.params[ha.indWo + e];
    for (int i = tid; i < 2 * NTMAX; i += 128) (&sV0[0][0])[i] = 0.f;
    for (int i = tid; i < 4 * NC; i += 128) (&sV1[0][0])[i] = 0.f;
    if (wv == 1 && lane < 32) { if (R.en < 8) S.sMisc[R.em][R.en] = R.hr.misc; if (R.en == 0) S.sActMsg[R.em] = R.hr.actMsg; }
    SSTAMP(1);
    vmDrain(); pairBarrier();
    SSTAMP(2);
    if (layer == 0 && lane < dS) sV0[0][lane] = sStates[lane];
    pairBarrier();
    float prevSt = 0.f;
    for (int it = 0; it <= nSteps; ++it) {
      SSTAMP(4 + it);
      if (layer == 0) {
        const int k = it;
        if (k < nSteps) {
          const int cb = k & 1;
          if (k + 1 < nSteps && lane < dS) sV0[cb ^ 1][lane] = sStates[(k + 1) * dS + lane];
          float blk = 0.f;
          lstm32LayerStep<NTMAX, true>(L, w, bias, sV0[cb], IN0, nIn, prevSt, wr, br, &sV0[cb ^ 1][IN0], blk, k <= T, (long long)b * a.K + k, lane,
                                       sAct + (0 * 17 + (k <= T ? k : 0)) * ACT);
          if (lane < NC) sV1[cb][lane] = blk;
        }
      } else {
        const int k = it - 1;
        if (k < 0) R.hr.hoist(ha, boundedMask, R.bpv, R.live, R.en);       // (this wavefront has no layer-step yet)
        else {
          const int cb = k & 1;
          float blk = 0.f;
          lstm32LayerStep<NTMAX, true>(L, w, bias, sV1[cb], NC, NC, prevSt, wr, br, &sV1[cb ^ 1][NC], blk, k <= T, (long long)b * a.K + k, lane,
                                       sAct + (1 * 17 + (k <= T ? k : 0)) * ACT);
          if (lane < NC) {
            if (k == T) { a.Yout[(size_t)b * a.ldY + lane] = blk; S.sYo[0][lane] = blk; }       // (memory: A operand of the output layer's weight gradient)
            if (k == T + 1) S.sYo[1][lane] = blk;
          }
        }
      }
      pairBarrier();
    }
  }
  SSTAMP(30);
  // ======================================================= backward: weight rows requested now ==================================
  const int j = 1 - wv;                                    // wavefront 0 = the top layer (one step ahead), wavefront 1 = layer 0
  const RecLayer L = a.L[j];
  f32x4 wq[NO / 4];
  {
    const f32x4* rw = reinterpret_cast<const f32x4*>(W + L.indW + (size_t)(j == 1 ? lane : nIn + (lane & 31)) * NO);
#pragma unroll
    for (int q = 0; q < NO / 4; ++q) wq[q] = rw[q];
  }
  const float wrB = (L.hasRes && lane < L.resW) ? W[L.indWr + lane] : 0.f;
  for (int k = T + 1; k < a.K; ++k) {      // rows of the steps this sample does not have: zero deltas
    const long long r = (long long)b * a.K + k;
    L.D[r * NO + lane] = 0.f; L.D[r * NO + lane + 64] = 0.f;
    if (L.hasRes && lane < NC) L.Rd[r * L.ldR + lane] = 0.f;
  }
  // ======================================================= output layer + head (wavefront 1) ====================================
  SSTAMP(31);
  if (wv == 1) stepHeadRun(R, ha, S, lane, b, slot, nextRow);
  if (wv == 1) SSTAMP1(32);
  vmDrain(); pairBarrier();
  SSTAMP(33);
  // ======================================================= back-propagation through time ========================================
  {
    const float dres = (j == 1 && lane < NC) ? S.sDres[lane] : 0.f;
    float nxtSt = 0.f, nxtF = 0.f;
    for (int it = 0; it <= T + 1; ++it) {
      const int k = T - it + (j == 1 ? 0 : 1);
      if (k >= 0 && k <= T) {
        const long long r = (long long)b * a.K + k;
        float res = 0.f;
        if (lane < NC) {
          const float eTop = j == 1 ? (k == T ? dres : 0.f) : sTop[k & 1][lane];
          const float* act = sAct + (j * 17 + k) * ACT;
          if (L.hasRes) { L.Rd[r * L.ldR + lane] = eTop; res = lane < L.resW ? eTop * wrB : 0.f; }
          const float D = eTop + (k < T ? sRec[j][lane] : 0.f);
          const float cellInpt = act[lane], IG = act[NC + lane], FG = act[2 * NC + lane], OG = act[3 * NC + lane], co = act[5 * NC + lane];
          const float prevSt = k > 0 ? (act - ACT)[4 * NC + lane] : 0.f;
          const float diff = (1.f - co * co) * D;
          const float sd = diff * OG + (k < T ? nxtSt * nxtF : 0.f);
          const float d0 = IG * sd;
          const float d1 = IG * (1.f - IG) * cellInpt * sd;
          const float d2 = k > 0 ? FG * (1.f - FG) * prevSt * sd : 0.f;
          const float d3 = OG * (1.f - OG) * D * co;
          sD[j][lane] = d0; sD[j][NC + lane] = d1; sD[j][2 * NC + lane] = d2; sD[j][3 * NC + lane] = d3;
          nxtSt = sd; nxtF = FG;
        }
        waveLdsSync();
        { const float u0 = sD[j][lane], u1 = sD[j][64 + lane]; L.D[r * NO + lane] = u0; L.D[r * NO + 64 + lane] = u1; }      // the row for the weight gradients: two whole-wavefront stores
        if (j == 1 || k > 0) {
          const f32x4* d4 = reinterpret_cast<const f32x4*>(sD[j]);
          f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f};
#pragma unroll
          for (int h0 = 0; h0 < NO / 4; h0 += NO / 8) {
            f32x4 dq[NO / 8];
#pragma unroll
            for (int q = 0; q < NO / 8; ++q) dq[q] = d4[h0 + q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NO / 8; q += 2) {
              const f32x4 da = dq[q], db = dq[q + 1];
              p0 += f32x2{wq[h0 + q][0], wq[h0 + q][1]} * f32x2{da[0], da[1]}; p1 += f32x2{wq[h0 + q][2], wq[h0 + q][3]} * f32x2{da[2], da[3]};
              p2 += f32x2{wq[h0 + q + 1][0], wq[h0 + q + 1][1]} * f32x2{db[0], db[1]}; p3 += f32x2{wq[h0 + q + 1][2], wq[h0 + q + 1][3]} * f32x2{db[2], db[3]};
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const f32x2 ps = (p0 + p1) + (p2 + p3);
          const float e = ps[0] + ps[1];
          if (j == 1) { if (lane < NC) sTop[k & 1][lane] = res + e; else sRec[1][lane - NC] = e; }
          else if (lane < NC) sRec[0][lane] = e;
        }
      }
      pairBarrier();
    }
  }
  SSTAMP(250);
}

// ---- MGU, two layers of 32 cells: the same arrangement ---------------------------------------------------------------------
// A layer-step has 64 gate columns -- forget gates on lanes 0..31, candidate states on lanes 32..63 (MGULayer::forward,
// Layer_GRU.h:64-124) -- and two dependent halves: the candidate's recurrent operand is prevOut * forget.  Lane o keeps column o
// of [W_in; W_rec] in registers; the forget lanes publish prevOut * f through LDS, the candidate lanes then add their 32
// recurrent terms.  Layer 0's wavefront runs one step ahead of layer 1's, as above.
template <int IN>
__device__ __forceinline__ float dotIn(const float (&w)[32], const float* vec) {
  const f32x4* v4 = reinterpret_cast<const f32x4*>(vec);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  f32x4 vq[IN / 4];      // (all reads before the first FMA: see lstm32LayerStep)
#pragma unroll
  for (int q = 0; q < IN / 4; ++q) vq[q] = v4[q];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < IN / 4; ++q) { const f32x4 v = vq[q]; a0 += w[4 * q] * v[0]; a1 += w[4 * q + 1] * v[1]; a2 += w[4 * q + 2] * v[2]; a3 += w[4 * q + 3] * v[3]; }
  return (a0 + a1) + (a2 + a3);
}

// (LDSACT: forget gate, candidate and output of the step go to `actRow` in LDS -- the one-launch step keeps them for its backward pass)
template <int IN, bool LDSACT = false>      // IN: operand length of the input part as the unrolled loop walks it (zero weights behind the layer's own inputs)
__device__ __forceinline__ void mgu32LayerStep(const RecLayer& L, const float (&win)[32], const float (&wrec)[32], float bias, const float* vec,
                                               float* hf, int nInL, float wr, float br, float* hNext, float& blkOut, bool store, long long r, int lane,
                                               float* actRow = nullptr) {
  constexpr int NC = 32, NO = 64;
  const float* hPrev = vec + IN;
  const float accIn = bias + dotIn<IN>(win, vec);
  const float po = lane < NC ? hPrev[lane] : 0.f;
  const float f = fastSigm(accIn + dotIn<NC>(wrec, hPrev));            // (meaningful on the forget lanes)
  if (lane < NC) hf[lane] = po * f;
  waveLdsSync();
  const float sc = fastTanh(accIn + dotIn<NC>(wrec, hf));              // (meaningful on the candidate lanes)
  if (store) {
    if (LDSACT) actRow[lane] = lane < NC ? f : sc; else L.X[r * NO + lane] = lane < NC ? f : sc;
    if (lane < nInL + NC) L.A[r * L.ldA + lane] = vec[lane < nInL ? lane : IN + (lane - nInL)];      // [input | previous output], one store
  }
  const float st = fromUpperHalf(sc);
  if (lane < NC) {
    const float out = f * st + (1.f - f) * po;                           // (po is 0 at the first step of the window)
    if (store) { if (LDSACT) actRow[NO + lane] = out; else L.Y[r * NO + lane] = out; L.A2[r * L.ldA2 + lane] = po * f; }
    float blk = out;
    if (L.hasRes && lane < L.resW) blk += vec[lane] * wr + br;
    hNext[lane] = out;
    blkOut = blk;
  }
}

template <int IN0>
__global__ __launch_bounds__(128) void mgu32_forward_wave_kernel(RecArgs a) {
  constexpr int NC = 32, NO = 64;
  __shared__ __attribute__((aligned(16))) float sV0[2][IN0 + NC];      // layer 0 operand [x_k | h0_{k-1}], double-buffered over the steps
  __shared__ __attribute__((aligned(16))) float sV1[2][2 * NC];        // layer 1 operand [block-0 output of step k | h1_{k-1}]
  __shared__ __attribute__((aligned(16))) float sHF[2][NC];            // prevOut * forget, per layer
  __shared__ float sStates[18 * 32];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int layer = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool acting = a.actStates != nullptr;                          // rollout inference: the agent's last states, nothing stored
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  const RecLayer L = a.L[layer];
  const int nIn = a.L[0].nIn, dS = a.dS;
  float win[32], wrec[32];
  {
    const float* Wl = W + L.indW;
    const int nInL = layer == 0 ? nIn : NC;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      win[i] = i < nInL ? Wl[(size_t)i * NO + lane] : 0.f;
      wrec[i] = Wl[(size_t)(nInL + i) * NO + lane];
    }
  }
  const float bias = W[L.indB + lane];
  const int c = lane & 31;
  float wr = 0.f, br = 0.f;
  if (L.hasRes && c < L.resW) { wr = W[L.indWr + c]; br = W[L.indBr + c]; }
  for (int e = tid; e < nSteps * dS; e += 128) {
    const int kk = e / dS, i = e - kk * dS;
    const float raw = acting ? a.actStates[e] : a.rp.S[(size_t)(slot - T + kk) * dS + i];
    sStates[e] = (raw - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  for (int i = tid; i < 2 * (IN0 + NC); i += 128) (&sV0[0][0])[i] = 0.f;
  for (int i = tid; i < 4 * NC; i += 128) (&sV1[0][0])[i] = 0.f;
  vmDrain(); pairBarrier();
  if (layer == 0 && lane < dS) sV0[0][lane] = sStates[lane];
  pairBarrier();
  for (int it = 0; it <= nSteps; ++it) {
    if (layer == 0) {
      const int k = it;
      if (k < nSteps) {
        const int cb = k & 1;
        if (k + 1 < nSteps && lane < dS) sV0[cb ^ 1][lane] = sStates[(k + 1) * dS + lane];
        float blk = 0.f;
        mgu32LayerStep<IN0>(L, win, wrec, bias, sV0[cb], sHF[0], nIn, wr, br, &sV0[cb ^ 1][IN0], blk, !acting && k <= T, (long long)b * a.K + k, lane);
        if (lane < NC) sV1[cb][lane] = blk;
      }
    } else {
      const int k = it - 1;
      if (k >= 0) {
        const int cb = k & 1;
        float blk = 0.f;
        mgu32LayerStep<NC>(L, win, wrec, bias, sV1[cb], sHF[1], NC, wr, br, &sV1[cb ^ 1][NC], blk, !acting && k <= T, (long long)b * a.K + k, lane);
        if (lane < NC) {
          if (k == T) a.Yout[(size_t)b * a.ldY + lane] = blk;
          if (k == T + 1) a.Yout[(size_t)nextRow * a.ldY + lane] = blk;
        }
      }
    }
    pairBarrier();
  }
}

// backward (MGULayer::backward, Layer_GRU.h:126-231): lane i holds ROW i of [W_in; W_rec] (32 forget + 32 candidate columns); the
// top layer's lanes 0..31 are its input rows, 32..63 its recurrent rows; layer 0 needs its recurrent rows only (lanes 0..31)
template <int IN0>
__global__ __launch_bounds__(128) void mgu32_backward_wave_kernel(RecArgs a) {
  constexpr int NC = 32, NO = 64, ACT = 3 * NC;            // per (step, layer): [forget | candidate | output]
  __shared__ __attribute__((aligned(16))) float sDS[2][NC], sDF[2][NC];
  __shared__ float sAct[2][17 * ACT];
  __shared__ float sTop[2][NC];
  __shared__ float sRec[2][NC];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = 1 - wv;                                    // wavefront 0 = the top layer (one step ahead), wavefront 1 = layer 0
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  const RecLayer L = a.L[j];
  const int nIn0 = a.L[0].nIn;
  float wf[32], ws[32];                                    // this lane's row: forget columns, candidate columns
  {
    const f32x4* rw = reinterpret_cast<const f32x4*>(W + L.indW + (size_t)(j == 1 ? lane : nIn0 + (lane & 31)) * NO);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f32x4 u = rw[q], v = rw[8 + q]; wf[4 * q] = u[0]; wf[4 * q + 1] = u[1]; wf[4 * q + 2] = u[2]; wf[4 * q + 3] = u[3];
                                  ws[4 * q] = v[0]; ws[4 * q + 1] = v[1]; ws[4 * q + 2] = v[2]; ws[4 * q + 3] = v[3]; }
  }
  {
    const int total = (T + 1) * ACT;
    for (int e = lane; e < total; e += 64) {
      const int k = e / ACT, x = e - k * ACT;
      const long long r = (long long)b * a.K + k;
      sAct[j][e] = x < NO ? L.X[r * NO + x] : L.Y[r * NO + (x - NO)];
    }
  }
  const float wr = (L.hasRes && lane < L.resW) ? W[L.indWr + lane] : 0.f;
  const float dres = (j == 1 && lane < NC) ? a.Dres[(size_t)b * a.ldD + lane] : 0.f;
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    L.D[r * NO + lane] = 0.f;
    if (L.hasRes && lane < NC) L.Rd[r * L.ldR + lane] = 0.f;
  }
  vmDrain(); pairBarrier();
  for (int it = 0; it <= T + 1; ++it) {
    const int k = T - it + (j == 1 ? 0 : 1);
    if (k >= 0 && k <= T) {
      const long long r = (long long)b * a.K + k;
      float res = 0.f, dLdO = 0.f, f = 0.f, sc = 0.f, po = 0.f, dS_ = 0.f;
      if (lane < NC) {
        const float eTop = j == 1 ? (k == T ? dres : 0.f) : sTop[k & 1][lane];
        const float* act = sAct[j] + k * ACT;
        if (L.hasRes) { L.Rd[r * L.ldR + lane] = eTop; res = lane < L.resW ? eTop * wr : 0.f; }
        dLdO = eTop + (k < T ? sRec[j][lane] : 0.f);
        f = act[lane]; sc = act[NC + lane]; po = k > 0 ? (act - ACT)[2 * NC + lane] : 0.f;
        dS_ = dLdO * f * (1.f - sc * sc);                                         // 1) dLdS
        sDS[j][lane] = dS_;
      }
      waveLdsSync();
      const float viaS = dotIn<NC>(ws, sDS[j]);                                    // row i: sum_o W[i][nC + o] dLdS[o]
      // 2) dLdFprevOut of cell c = the recurrent row nIn + c
      float fp = j == 1 ? fromUpperHalf(viaS) : viaS;
      if (k == 0) fp = 0.f;
      if (lane < NC) {
        const float dF = ((sc - po) * dLdO + fp * po) * f * (1.f - f);             // 3) dLdF
        sDF[j][lane] = dF;
        L.D[r * NO + lane] = dF; L.D[r * NO + NC + lane] = dS_;
      }
      waveLdsSync();
      const float viaF = dotIn<NC>(wf, sDF[j]);                                    // row i: sum_o W[i][o] dLdF[o]
      const float g = j == 1 ? fromUpperHalf(viaF) : viaF;
      if (lane < NC) {
        if (j == 1) sTop[k & 1][lane] = (res + viaF) + viaS;                       // error of block 0's output (+ the residual path)
        if (k > 0) sRec[j][lane] = ((1.f - f) * dLdO + f * fp) + g;                // 4) dLdprevOut
      }
    }
    pairBarrier();
  }
}

// ---- the MGU form of the one-launch step (see lstm32_step_wave_kernel): window forward, output layer + head, back-propagation through
// time of a sample in one workgroup; forget gates, candidates and outputs of the window stay in LDS ---------------------------------
template <int IN0>
__global__ __launch_bounds__(256) void mgu32_step_wave_kernel(RecArgs a, HeadArgs ha, unsigned long long boundedMask, ExtraArgs extra) {
  constexpr int NC = 32, NO = 64, ACT = 3 * NC;            // per (step, layer): [forget | candidate | output]
  constexpr int O_V0 = 2 * 17 * ACT * 4, O_V1 = O_V0 + 2 * (IN0 + NC) * 4, O_HF = O_V1 + 2 * 64 * 4, O_ST = O_HF + 2 * NC * 4, O_DS = O_ST + 18 * 32 * 4,
                O_DF = O_DS + 2 * NC * 4, O_TOP = O_DF + 2 * NC * 4, O_REC = O_TOP + 2 * NC * 4, O_HEAD = O_REC + 2 * NC * 4, TOTAL = O_HEAD + STEP_HEAD_LDS;
  static_assert(O_HEAD % 16 == 0 && O_V0 % 16 == 0 && O_V1 % 16 == 0 && O_HF % 16 == 0 && O_DS % 16 == 0 && O_DF % 16 == 0, "alignment of the 16-byte reads");
  __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL > (int)TAIL_LDS_BYTES ? TOTAL : (int)TAIL_LDS_BYTES];
  if (extra.role) { if (blockIdx.x == 0) { runExtra(extra, smem); return; } if (threadIdx.x >= 128) return; }
  float* sAct = reinterpret_cast<float*>(smem);                                   // [2][17][ACT]
  float (*sV0)[IN0 + NC] = reinterpret_cast<float (*)[IN0 + NC]>(smem + O_V0);
  float (*sV1)[2 * NC] = reinterpret_cast<float (*)[2 * NC]>(smem + O_V1);
  float (*sHF)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_HF);
  float* sStates = reinterpret_cast<float*>(smem + O_ST);
  float (*sDS)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_DS);
  float (*sDF)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_DF);
  float (*sTop)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_TOP);
  float (*sRec)[NC] = reinterpret_cast<float (*)[NC]>(smem + O_REC);
  const StepHeadLds S = stepHeadLds(smem + O_HEAD);

  const int b = blockIdx.x - (extra.role ? 1 : 0), tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = a.bt.t[b]; const long long slot = a.bt.slot[b];
  const int T = min(a.nBPTT, t);
  const int nextRow = a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  const int nIn = a.L[0].nIn, dS = a.dS;
  StepHeadRegs R;
  stepHeadLoad(R, ha, a.sc, wv, lane, slot, nextRow);
  {
    // ================================================ forward over the window ==================================================
    const int layer = wv;
    const RecLayer L = a.L[layer];
    float win[32], wrec[32];
    {
      const float* Wl = W + L.indW;
      const int nInL = layer == 0 ? nIn : NC;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        win[i] = i < nInL ? Wl[(size_t)i * NO + lane] : 0.f;
        wrec[i] = Wl[(size_t)(nInL + i) * NO + lane];
      }
    }
    const float bias = W[L.indB + lane];
    const int c = lane & 31;
    float wr = 0.f, br = 0.f;
    if (L.hasRes && c < L.resW) { wr = W[L.indWr + c]; br = W[L.indBr + c]; }
    for (int e = tid; e < nSteps * dS; e += 128) {
      const int kk = e / dS, i = e - kk * dS;
      sStates[e] = (a.rp.S[(size_t)(slot - T + kk) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
    }
    for (int e = tid; e < NC * ha.ldWo; e += 128) S.sWo[e] = ha.params[ha.indWo + e];
    for (int i = tid; i < 2 * (IN0 + NC); i += 128) (&sV0[0][0])[i] = 0.f;
    for (int i = tid; i < 4 * NC; i += 128) (&sV1[0][0])[i] = 0.f;
    if (wv == 1 && lane < 32) { if (R.en < 8) S.sMisc[R.em][R.en] = R.hr.misc; if (R.en == 0) S.sActMsg[R.em] = R.hr.actMsg; }
    vmDrain(); pairBarrier();
    if (layer == 0 && lane < dS) sV0[0][lane] = sStates[lane];
    pairBarrier();
    for (int it = 0; it <= nSteps; ++it) {
      if (layer == 0) {
        const int k = it;
        if (k < nSteps) {
          const int cb = k & 1;
          if (k + 1 < nSteps && lane < dS) sV0[cb ^ 1][lane] = sStates[(k + 1) * dS + lane];
          float blk = 0.f;
          mgu32LayerStep<IN0, true>(L, win, wrec, bias, sV0[cb], sHF[0], nIn, wr, br, &sV0[cb ^ 1][IN0], blk, k <= T, (long long)b * a.K + k, lane,
                                    sAct + (0 * 17 + (k <= T ? k : 0)) * ACT);
          if (lane < NC) sV1[cb][lane] = blk;
        }
      } else {
        const int k = it - 1;
        if (k < 0) R.hr.hoist(ha, boundedMask, R.bpv, R.live, R.en);       // (this wavefront has no layer-step yet)
        else {
          const int cb = k & 1;
          float blk = 0.f;
          mgu32LayerStep<NC, true>(L, win, wrec, bias, sV1[cb], sHF[1], NC, wr, br, &sV1[cb ^ 1][NC], blk, k <= T, (long long)b * a.K + k, lane,
                                   sAct + (1 * 17 + (k <= T ? k : 0)) * ACT);
          if (lane < NC) {
            if (k == T) { a.Yout[(size_t)b * a.ldY + lane] = blk; S.sYo[0][lane] = blk; }
            if (k == T + 1) S.sYo[1][lane] = blk;
          }
        }
      }
      pairBarrier();
    }
  }
  // ======================================================= backward: weight rows requested now ==================================
  const int j = 1 - wv;                                    // wavefront 0 = the top layer (one step ahead), wavefront 1 = layer 0
  const RecLayer L = a.L[j];
  float wf[32], ws[32];                                    // this lane's row: forget columns, candidate columns
  {
    const f32x4* rw = reinterpret_cast<const f32x4*>(W + L.indW + (size_t)(j == 1 ? lane : nIn + (lane & 31)) * NO);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f32x4 u = rw[q], v = rw[8 + q]; wf[4 * q] = u[0]; wf[4 * q + 1] = u[1]; wf[4 * q + 2] = u[2]; wf[4 * q + 3] = u[3];
                                  ws[4 * q] = v[0]; ws[4 * q + 1] = v[1]; ws[4 * q + 2] = v[2]; ws[4 * q + 3] = v[3]; }
  }
  const float wrB = (L.hasRes && lane < L.resW) ? W[L.indWr + lane] : 0.f;
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    L.D[r * NO + lane] = 0.f;
    if (L.hasRes && lane < NC) L.Rd[r * L.ldR + lane] = 0.f;
  }
  // ======================================================= output layer + head (wavefront 1) ====================================
  if (wv == 1) stepHeadRun(R, ha, S, lane, b, slot, nextRow);
  vmDrain(); pairBarrier();
  // ======================================================= back-propagation through time ========================================
  {
    const float dres = (j == 1 && lane < NC) ? S.sDres[lane] : 0.f;
    for (int it = 0; it <= T + 1; ++it) {
      const int k = T - it + (j == 1 ? 0 : 1);
      if (k >= 0 && k <= T) {
        const long long r = (long long)b * a.K + k;
        float res = 0.f, dLdO = 0.f, f = 0.f, sc = 0.f, po = 0.f, dS_ = 0.f;
        if (lane < NC) {
          const float eTop = j == 1 ? (k == T ? dres : 0.f) : sTop[k & 1][lane];
          const float* act = sAct + (j * 17 + k) * ACT;
          if (L.hasRes) { L.Rd[r * L.ldR + lane] = eTop; res = lane < L.resW ? eTop * wrB : 0.f; }
          dLdO = eTop + (k < T ? sRec[j][lane] : 0.f);
          f = act[lane]; sc = act[NC + lane]; po = k > 0 ? (act - ACT)[2 * NC + lane] : 0.f;
          dS_ = dLdO * f * (1.f - sc * sc);                                         // 1) dLdS
          sDS[j][lane] = dS_;
        }
        waveLdsSync();
        const float viaS = dotIn<NC>(ws, sDS[j]);                                    // row i: sum_o W[i][nC + o] dLdS[o]
        float fp = j == 1 ? fromUpperHalf(viaS) : viaS;                              // 2) dLdFprevOut of cell c = the recurrent row nIn + c
        if (k == 0) fp = 0.f;
        if (lane < NC) {
          const float dF = ((sc - po) * dLdO + fp * po) * f * (1.f - f);             // 3) dLdF
          sDF[j][lane] = dF;
        }
        waveLdsSync();
        L.D[r * NO + lane] = lane < NC ? sDF[j][lane] : sDS[j][lane - NC];          // the row for the weight gradients [dLdF | dLdS]: one whole-wavefront store
        const float viaF = dotIn<NC>(wf, sDF[j]);                                    // row i: sum_o W[i][o] dLdF[o]
        const float g = j == 1 ? fromUpperHalf(viaF) : viaF;
        if (lane < NC) {
          if (j == 1) sTop[k & 1][lane] = (res + viaF) + viaS;                       // error of block 0's output (+ the residual path)
          if (k > 0) sRec[j][lane] = ((1.f - f) * dLdO + f * fp) + g;                // 4) dLdprevOut
        }
      }
      pairBarrier();
    }
  }
}

// ---- nnType "RNN": dense layers with a recurrent term (BaseLayer with bRecurrent; Network/Builder.cpp:76-81,
// Network/Layers/Layer_Base.h:64-113): x_t = W_in^T in_t + W_rec^T y_{t-1} + b, y_t = f(x_t); weights [W_in; W_rec] row-major with the
// dense layers' pitch roundUp(cells, 8); backward Layer::backward with NR = cells (Layers.h:123-188).  One workgroup per sample
// walks the window; the weights of all layers sit in LDS (pitch + 1: the forward pass reads columns, the backward pass rows);
// a cell's sum is split over the four wavefronts (rows i = wave, wave + 4, ...) and joined in wave order.  Rows kept per
// (sample, step): [input | previous output] (A operand of the weight gradients), x, y, and the deltas after f'. ----
__device__ __forceinline__ int rnnPitch(int nC) { return ((nC + 7) & ~7) + 1; }
__device__ __forceinline__ void rnnStageWeights(const RecArgs& a, float* sW, int tid) {
  int off = 0;
  for (int j = 0; j < a.nL; ++j) {
    const RecLayer& L = a.L[j];
    const int ld = (L.nC + 7) & ~7, rows = L.nIn + L.nC, pitch = ld + 1;
    const float* src = a.W + L.indW;
    const int total = rows * ld;
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; v[u] = e < total ? src[e] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < total) { const int i = e / ld, o = e - i * ld; sW[off + i * pitch + o] = v[u]; } }
    }
    off += rows * pitch;
  }
}
__device__ __forceinline__ int rnnLdsOffset(const RecArgs& a, int j) {
  int off = 0;
  for (int q = 0; q < j; ++q) off += (a.L[q].nIn + a.L[q].nC) * rnnPitch(a.L[q].nC);
  return off;
}
template <bool LDSW>
__global__ __launch_bounds__(256) void rnn_forward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sBuf[2][REC_GENIN];
  __shared__ float sPrevOut[HL_MAX_HIDDEN][REC_GENC];
  __shared__ float sPart[4][REC_GENC];
  const int b = blockIdx.x, tid = threadIdx.x, c0 = tid & 63, part = tid >> 6;
  const bool acting = a.actStates != nullptr;
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  if constexpr (LDSW) rnnStageWeights(a, sW, tid);
  float bias[HL_MAX_HIDDEN], wr[HL_MAX_HIDDEN], br[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) {
    bias[j] = 0.f; wr[j] = 0.f; br[j] = 0.f;
    if (j < a.nL) {
      const RecLayer& L = a.L[j];
      if (tid < L.nC) bias[j] = W[L.indB + tid];
      if (L.hasRes && tid < L.resW) { wr[j] = W[L.indWr + tid]; br[j] = W[L.indBr + tid]; }
    }
  }
  const float sMean = tid < a.dS ? a.rp.stMean[tid] : 0.f, sScale = tid < a.dS ? a.rp.stScale[tid] : 1.f;
  vmDrain(); ldsBarrier();
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    const long long sl = slot - T + k;
    if (a.Xin != nullptr || a.nApp > 0) { for (int e = tid; e < a.L[0].nIn; e += 256) sBuf[0][e] = recInputAt(a, acting, b, slot, t, T, nextRow, k, e); }
    else if (tid < a.dS) { const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)sl * a.dS + tid]; sBuf[0][tid] = (raw - sMean) * sScale; }
    ldsBarrier();
    int cur = 0;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      const int nIn = L.nIn, nC = L.nC, ld = (nC + 7) & ~7;
      const float* in = sBuf[cur];
      const int pitch = LDSW ? ld + 1 : ld, wOff = LDSW ? rnnLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * pitch + o]; else return gWj[(size_t)i * pitch + o]; };
      if (store) {
        for (int i = tid; i < nIn; i += 256) L.A[r * L.ldA + i] = in[i];
        if (tid < nC) L.A[r * L.ldA + nIn + tid] = k > 0 ? sPrevOut[j][tid] : 0.f;
      }
      for (int c = c0; c < nC; c += 64) {       // quarter sums over the rows i = part, part + 4, ...: first the inputs, then the previous outputs
        float acc = 0.f;
#pragma unroll 4
        for (int i = part; i < nIn; i += 4) acc += in[i] * wAt(i, c);
        if (k > 0) {
#pragma unroll 4
          for (int i = part; i < nC; i += 4) acc += sPrevOut[j][i] * wAt(nIn + i, c);
        }
        sPart[part][c] = acc;
      }
      ldsBarrier();
      float out = 0.f;
      if (tid < nC) {
        const float x = bias[j] + ((sPart[0][tid] + sPart[1][tid]) + (sPart[2][tid] + sPart[3][tid]));
        out = actEval(a.func, x);
        if (store) { L.X[r * nC + tid] = x; L.Y[r * nC + tid] = out; }
        float blk = out;                                   // ParametricResidualLayer::forward (Layers.h:347-361)
        if (L.hasRes && tid < L.resW) blk += in[tid] * wr[j] + br[j];
        sBuf[cur ^ 1][tid] = blk;
      }
      ldsBarrier();
      if (tid < nC) sPrevOut[j][tid] = out;
      cur ^= 1;
    }
    const int nCl = a.L[a.nL - 1].nC;
    if (a.YoutRows) {      // lower segment of a two-type stack: every step's output is an input row of the segment above
      const long long ro = k <= T ? (long long)b * a.K + k : (long long)a.B * a.K + (nextRow - a.B);
      if (tid < nCl) a.YoutRows[ro * a.ldYR + tid] = sBuf[cur][tid];
    } else {
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    }
    ldsBarrier();
  }
}
template <bool LDSW>
__global__ __launch_bounds__(256) void rnn_backward_kernel(RecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  __shared__ float sTop[2][REC_GENIN];                    // error w.r.t. the output of the current block (from above, same step)
  __shared__ float sRec[HL_MAX_HIDDEN][REC_GENC];          // error w.r.t. this step's output coming from step k+1
  __shared__ float sD[REC_GENC], sRes[REC_GENC], sRecNew[REC_GENC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  if constexpr (LDSW) rnnStageWeights(a, sW, tid);
  float wr[HL_MAX_HIDDEN];
#pragma unroll
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) { wr[j] = 0.f; if (j < a.nL && a.L[j].hasRes && tid < a.L[j].resW) wr[j] = W[a.L[j].indWr + tid]; }
  vmDrain(); ldsBarrier();
  for (int k = T + 1; k < a.K; ++k) {      // rows of the steps this sample does not have: zero deltas
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      if (tid < L.nC) { L.D[r * L.nC + tid] = 0.f; if (L.hasRes) L.Rd[r * L.ldR + tid] = 0.f; }
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    const int nCl = a.L[a.nL - 1].nC;
    if (tid < nCl) sTop[0][tid] = a.DresRows ? a.DresRows[r * a.ldDR + tid] : (k == T ? a.Dres[(size_t)b * a.ldD + tid] : 0.f);
    ldsBarrier();
    for (int j = a.nL - 1; j >= 0; --j) {
      const RecLayer& L = a.L[j];
      const int nIn = L.nIn, nC = L.nC, ld = (nC + 7) & ~7;
      const int pitch = LDSW ? ld + 1 : ld, wOff = LDSW ? rnnLdsOffset(a, j) : 0;
      const float* gWj = W + L.indW;
      auto wAt = [&](int i, int o) -> float { if constexpr (LDSW) return sW[wOff + i * pitch + o]; else return gWj[(size_t)i * pitch + o]; };
      if (tid < nC) {
        const float eTop = sTop[cur][tid];
        if (L.hasRes) { L.Rd[r * L.ldR + tid] = eTop; sRes[tid] = tid < L.resW ? eTop * wr[j] : 0.f; }
        const float D = eTop + (k < T ? sRec[j][tid] : 0.f);
        const float d = D * actDiff(a.func, L.X[r * nC + tid], L.Y[r * nC + tid]);     // BaseLayer::backward (Layer_Base.h:97-113)
        sD[tid] = d; L.D[r * nC + tid] = d;
      }
      ldsBarrier();
      // Layer::backward (Layers.h:123-188): rows 0..nIn-1 of [W_in; W_rec] give the error of the block below (not below the first
      // layer), rows nIn.. the error handed to the previous step; one row per group of four lanes, quarter sums joined by shuffles
      {
        const int part = tid & 3, row0 = j > 0 ? 0 : nIn, nRow = nIn + (k > 0 ? nC : 0);
        for (int i0 = row0; i0 < nRow; i0 += 64) {
          const int i = i0 + (tid >> 2);
          float e = 0.f;
          if (i < nRow) for (int o = part; o < nC; o += 4) e += wAt(i, o) * sD[o];
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
// Convolutional layers in front of recurrent ones: every step of a sample's window passes through the conv stack, so the conv
// launches run over B K window rows (+ the truncated next states behind them) instead of B sampled rows.  This kernel writes the
// row -> (replay slot, step) map in the form stack_gather_kernel / the conv kernels already read: rows r = b K + k < B K hold step
// t - T + min(k, T) of sample b (rows of steps a window does not have repeat its last one: their deltas are zero), next row j
// belongs to window row b K + T; the row count goes into a DevScalars of its own.
__global__ __launch_bounds__(256) void window_rows_kernel(WinRowsArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nNext = a.sc->nRows[a.parity] - a.B, BK = a.B * a.K;
  if (i == 0) { a.scW->nRows[a.parity] = BK + nNext; a.scW->nNext[a.parity] = nNext; }
  if (i < BK) {
    const int b = i / a.K, k = i - b * a.K, t = a.t[b], T = min(a.nBPTT, t), kk = min(k, T);
    a.slotW[i] = a.slot[b] - T + kk; a.tW[i] = t - T + kk;
  } else if (i - BK < nNext) {
    const int j = i - BK, b = a.nextSrc[j];
    a.nextSrcW[j] = b * a.K + min(a.nBPTT, a.t[b]);
  }
}
hipError_t launch_window_rows(const WinRowsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(window_rows_kernel, dim3((a.B * a.K + a.B + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

static size_t rnnLdsBytes(const RecArgs& a) {
  size_t fl = 0;
  for (int j = 0; j < a.nL; ++j) fl += (size_t)(a.L[j].nIn + a.L[j].nC) * (((a.L[j].nC + 7) & ~7) + 1);
  return fl * sizeof(float);
}

static bool mgu32Wave(const RecArgs& a) {
  return a.gates == 2 && (a.actStates == nullptr || a.actSteps <= 17) && a.nL == 2 && a.L[0].nC == 32 && a.L[1].nC == 32 && a.L[1].nIn == 32 &&
         a.L[0].nIn <= 32 && a.dS == a.L[0].nIn && a.K <= 17 && a.L[0].indW % 4 == 0 && a.L[1].indW % 4 == 0 && !a.L[0].hasRes;
}
// the wave-per-sample kernels serve the training pass of two LSTM layers of 32 cells each over up to 32 inputs and 17 steps
static bool lstm32Wave(const RecArgs& a) {      // (forward: training windows and rollout inference; backward: training only)
  return a.gates == 4 && (a.actStates == nullptr || a.actSteps <= 17) && a.nL == 2 && a.L[0].nC == 32 && a.L[1].nC == 32 && a.L[1].nIn == 32 &&
         a.L[0].nIn <= 32 && a.dS == a.L[0].nIn && a.K <= 17 && a.L[0].indW % 4 == 0 && a.L[1].indW % 4 == 0 && !a.L[0].hasRes;
}
// shapes only the one-gate-per-thread kernels serve: inputs other than the step's own observed state, layers wider than 64 cells
static bool recGeneral(const RecArgs& a) {
  bool wide = false;
  for (int j = 0; j < a.nL; ++j) wide = wide || a.L[j].nC > REC_MAXC || a.L[j].nIn > REC_MAXIN;
  return wide || a.Xin != nullptr || a.nApp > 0;
}
static size_t recLdsBytes(const RecArgs& a) {
  size_t fl = 0;
  for (int j = 0; j < a.nL; ++j) fl += (size_t)(a.L[j].nIn + a.L[j].nC) * (a.gates * a.L[j].nC + 1);
  return fl * sizeof(float);
}
template <class K> static hipError_t recLaunch(K kernel, const RecArgs& a, size_t lds, size_t*, hipStream_t s) {
  if (lds > 0) { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(kernel), lds); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL(kernel, dim3(a.B), dim3(256), lds, s, a);
  return hipGetLastError();
}
// (static LDS of the kernels comes on top of the weights; 160 KB per workgroup)
hipError_t launch_rec_forward(const RecArgs& a, hipStream_t s) {
  if (rec_tm_ok(a)) return launch_rec_tm_forward(a, s);      // wide LSTM layers, training windows: time-step-major on the MFMA (rectm.hip)
  if (rec_tm_act_ok(a)) return launch_rec_tm_forward(a, s);  // ... and the acting window of nets wider than the kernels below hold
  // (static LDS of the general kernels: up to 46 KB)
  static size_t attr[4] = {0, 0, 0, 0}; const size_t lds = recLdsBytes(a); const bool fit = lds <= 100 * 1024; const bool general = recGeneral(a);
  if (a.gates == 1) { const size_t l1 = rnnLdsBytes(a); return l1 <= 100 * 1024 ? recLaunch(rnn_forward_kernel<true>, a, l1, &attr[0], s) : recLaunch(rnn_forward_kernel<false>, a, 0, &attr[1], s); }
  if (general) {
    if (a.gates == 2) return fit ? recLaunch(mgu_forward_kernel<true, 0>, a, lds, &attr[0], s) : recLaunch(mgu_forward_kernel<false, 0>, a, 0, &attr[1], s);
    return fit ? recLaunch(rec_forward_kernel<true>, a, lds, &attr[2], s) : recLaunch(rec_forward_kernel<false>, a, 0, &attr[3], s);
  }
  if (mgu32Wave(a)) {        // two layers of 32 cells, training pass: one wavefront per (sample, layer), weights in registers
    const int in0 = (a.L[0].nIn + 3) & ~3;
    if (in0 <= 4) hipLaunchKernelGGL(mgu32_forward_wave_kernel<4>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 8) hipLaunchKernelGGL(mgu32_forward_wave_kernel<8>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 16) hipLaunchKernelGGL(mgu32_forward_wave_kernel<16>, dim3(a.B), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(mgu32_forward_wave_kernel<32>, dim3(a.B), dim3(128), 0, s, a);
    return hipGetLastError();
  }
  if (a.gates == 2) {
    static size_t attrM[4] = {0, 0, 0, 0};
    if (fit && a.nL == 1) return recLaunch(mgu_forward_kernel<true, 1>, a, lds, &attrM[1], s);
    if (fit && a.nL == 2) return recLaunch(mgu_forward_kernel<true, 2>, a, lds, &attrM[2], s);
    return fit ? recLaunch(mgu_forward_kernel<true, 0>, a, lds, &attr[0], s) : recLaunch(mgu_forward_kernel<false, 0>, a, 0, &attr[1], s);
  }
  if (lstm32Wave(a)) {      // two layers of 32 cells, training pass: one wavefront per sample, weights in registers
    const int in0 = (a.L[0].nIn + 3) & ~3;
    if (in0 <= 4) hipLaunchKernelGGL(lstm32_forward_wave_kernel<4>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 8) hipLaunchKernelGGL(lstm32_forward_wave_kernel<8>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 16) hipLaunchKernelGGL(lstm32_forward_wave_kernel<16>, dim3(a.B), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(lstm32_forward_wave_kernel<32>, dim3(a.B), dim3(128), 0, s, a);
    return hipGetLastError();
  }
  size_t fl = 0; bool al = true;
  for (int j = 0; j < a.nL; ++j) { fl += (size_t)4 * a.L[j].nC * lstmGeo(a.L[j].nIn, a.L[j].nC).ld; al = al && a.L[j].indW % 4 == 0 && a.L[j].nIn <= REC_MAXIN; }
  if (al && fl * sizeof(float) <= 120 * 1024) {
    // (hidden layers above the first take the block below as input; the specialised bodies rely on nIn == cells there)
    bool same = true;
    for (int j = 0; j < a.nL; ++j) same = same && a.L[j].nC == a.L[0].nC && (j == 0 || a.L[j].nIn == a.L[0].nC);
    static size_t attrS[4] = {0, 0, 0, 0};
    if (same && a.nL == 2 && a.L[0].nC == 32) return recLaunch(lstm_forward_lds_kernel<2, 32>, a, fl * sizeof(float), &attrS[0], s);   // RACER_RNN.json
    if (a.nL == 1) return recLaunch(lstm_forward_lds_kernel<1, 0>, a, fl * sizeof(float), &attrS[1], s);
    if (a.nL == 2) return recLaunch(lstm_forward_lds_kernel<2, 0>, a, fl * sizeof(float), &attrS[2], s);
    if (a.nL == 3) return recLaunch(lstm_forward_lds_kernel<3, 0>, a, fl * sizeof(float), &attrS[3], s);
    return recLaunch(lstm_forward_lds_kernel<0, 0>, a, fl * sizeof(float), &attr[2], s);
  }
  return recLaunch(rec_forward_kernel<false>, a, 0, &attr[3], s);
}
// the one-launch step (lstm32_step_wave_kernel): the shapes of the wave-per-(sample, layer) LSTM kernels with a head of up to 32
// dense outputs and 16 action components / options
bool rec_step_fused_ok(const RecArgs& a, const HeadArgs& ha) {
  const int comps = ha.nOpt ? ha.nOpt : ha.dA;
  return !recGeneral(a) && (lstm32Wave(a) || mgu32Wave(a)) && a.actStates == nullptr && a.YoutRows == nullptr && a.DresRows == nullptr && a.K <= 17 &&
         ha.H == 32 && ha.nDense <= 32 && ha.nOut <= 48 && comps <= 16 && ha.ldWo <= 40;
}
hipError_t launch_rec_step_fused(const RecArgs& a, const HeadArgs& ha, const ExtraArgs* extra, hipStream_t s) {
  if (!rec_step_fused_ok(a, ha)) return hipErrorInvalidValue;
  ExtraArgs ex{}; if (extra) ex = *extra;
  unsigned long long mask = 0; for (int c = 0; c < HL_MAX_DIMA && c < 64; ++c) if (ha.bounded[c]) mask |= 1ull << c;
  const dim3 grid(a.B + (ex.role ? 1 : 0)), block(ex.role ? 256 : 128);
  const int in0 = (a.L[0].nIn + 3) & ~3;
  if (a.gates == 2) {
    if (in0 <= 4) hipLaunchKernelGGL(mgu32_step_wave_kernel<4>, grid, block, 0, s, a, ha, mask, ex);
    else if (in0 <= 8) hipLaunchKernelGGL(mgu32_step_wave_kernel<8>, grid, block, 0, s, a, ha, mask, ex);
    else if (in0 <= 16) hipLaunchKernelGGL(mgu32_step_wave_kernel<16>, grid, block, 0, s, a, ha, mask, ex);
    else hipLaunchKernelGGL(mgu32_step_wave_kernel<32>, grid, block, 0, s, a, ha, mask, ex);
    return hipGetLastError();
  }
  if (in0 <= 4) hipLaunchKernelGGL(lstm32_step_wave_kernel<4>, grid, block, 0, s, a, ha, mask, ex);
  else if (in0 <= 8) hipLaunchKernelGGL(lstm32_step_wave_kernel<8>, grid, block, 0, s, a, ha, mask, ex);
  else if (in0 <= 16) hipLaunchKernelGGL(lstm32_step_wave_kernel<16>, grid, block, 0, s, a, ha, mask, ex);
  else hipLaunchKernelGGL(lstm32_step_wave_kernel<32>, grid, block, 0, s, a, ha, mask, ex);
  return hipGetLastError();
}
hipError_t launch_rec_backward(const RecArgs& a, hipStream_t s) {
  if (rec_tm_ok(a)) return launch_rec_tm_backward(a, s);
  static size_t attr[4] = {0, 0, 0, 0}; const size_t lds = recLdsBytes(a); const bool fit = lds <= 100 * 1024; const bool general = recGeneral(a);
  if (a.gates == 1) { const size_t l1 = rnnLdsBytes(a); return l1 <= 100 * 1024 ? recLaunch(rnn_backward_kernel<true>, a, l1, &attr[0], s) : recLaunch(rnn_backward_kernel<false>, a, 0, &attr[1], s); }
  if (general) {
    if (a.gates == 2) return fit ? recLaunch(mgu_backward_kernel<true, 0>, a, lds, &attr[0], s) : recLaunch(mgu_backward_kernel<false, 0>, a, 0, &attr[1], s);
    return fit ? recLaunch(rec_backward_kernel<true>, a, lds, &attr[2], s) : recLaunch(rec_backward_kernel<false>, a, 0, &attr[3], s);
  }
  if (a.gates == 2) {
    static size_t attrM[4] = {0, 0, 0, 0};
    if (mgu32Wave(a)) { hipLaunchKernelGGL(mgu32_backward_wave_kernel<32>, dim3(a.B), dim3(128), 0, s, a); return hipGetLastError(); }
    if (fit && a.nL == 1) return recLaunch(mgu_backward_kernel<true, 1>, a, lds, &attrM[1], s);
    if (fit && a.nL == 2) return recLaunch(mgu_backward_kernel<true, 2>, a, lds, &attrM[2], s);
    return fit ? recLaunch(mgu_backward_kernel<true, 0>, a, lds, &attr[0], s) : recLaunch(mgu_backward_kernel<false, 0>, a, 0, &attr[1], s);
  }
  if (lstm32Wave(a)) {
    const int in0 = (a.L[0].nIn + 3) & ~3;
    if (in0 <= 4) hipLaunchKernelGGL(lstm32_backward_wave_kernel<4>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 8) hipLaunchKernelGGL(lstm32_backward_wave_kernel<8>, dim3(a.B), dim3(128), 0, s, a);
    else if (in0 <= 16) hipLaunchKernelGGL(lstm32_backward_wave_kernel<16>, dim3(a.B), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(lstm32_backward_wave_kernel<32>, dim3(a.B), dim3(128), 0, s, a);
    return hipGetLastError();
  }
  size_t fl = 0; bool al = true;
  for (int j = 0; j < a.nL; ++j) {
    fl += (size_t)(a.L[j].nIn + a.L[j].nC) * lstmBwdPitch(4 * a.L[j].nC) + (size_t)a.K * 6 * a.L[j].nC;
    al = al && a.L[j].indW % 4 == 0 && a.L[j].nIn <= REC_MAXIN;
  }
  if (al && fl * sizeof(float) <= 120 * 1024) {
    static size_t attrS[4] = {0, 0, 0, 0};
    if (a.nL == 1) return recLaunch(lstm_backward_lds_kernel<1>, a, fl * sizeof(float), &attrS[1], s);
    if (a.nL == 2) return recLaunch(lstm_backward_lds_kernel<2>, a, fl * sizeof(float), &attrS[2], s);
    if (a.nL == 3) return recLaunch(lstm_backward_lds_kernel<3>, a, fl * sizeof(float), &attrS[3], s);
    return recLaunch(lstm_backward_lds_kernel<0>, a, fl * sizeof(float), &attrS[0], s);
  }
  return fit ? recLaunch(rec_backward_kernel<true>, a, lds, &attr[2], s) : recLaunch(rec_backward_kernel<false>, a, 0, &attr[3], s);
}

}  // namespace hl

#ifdef REC_STAMPS
extern "C" int hl_debug_rec_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hl::recStamps), sizeof(unsigned long long) * 256);
}
#endif
