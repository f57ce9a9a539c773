// smarties_amd/csrc/fused.hip -- forward + V-RACER head + input-gradient back-propagation of the
// two-hidden-layer MLP as ONE kernel (the cfg-NS class of networks: dS <= 32, two hidden blocks of
// equal width H in {16,32,64,128,256}, dA <= 7).
//
// Why: a gradient step is a chain of dependent kernels, each costing a dispatch gap (~1.6 us), a
// cold L2 (every kernel boundary invalidates it) and a prologue.  fwd0 -> fwd1 -> head -> dX were
// four such kernels for ~40 MFLOP.  Here a 16-row PANEL of the minibatch is owned by a group of
// HT = H/16 workgroups (one 16-column tile each, all on one XCD because workgroups are dealt
// round-robin to the 8 XCDs and the group shares blockIdx % 8); inside the group only ONE exchange
// is needed:
//
//   every WG:  h1 = f(S W0 + b0) for the whole panel (17x256 weights: recomputed, not exchanged)
//              its tile of x2 = h1 W1 + b1, y3 = f(x2) + w*h1 + b and f'(x2)
//              -> global (agent-scope write-through stores)
//   -- group barrier: a counter per panel, one atomic arrive per WG, no L2 flush / invalidate --
//   every WG:  reads the panel's y3 and f'(x2) back (L2 hits), output layer + V-RACER head for the
//              16 samples (fp64, one (sample, action dim) per lane), delta_y3 = delta_out Wout^T,
//              delta_x2 = delta_y3 f'(x2) for the whole panel, its tile of
//              delta_h1 = delta_x2 W1^T (+ residual path), delta_x1 = delta_h1 f'(x1)
//
// The weight-gradient GEMMs need all rows and stay a second kernel (gemm16.hip, role DW).
// Reference functions: BaseLayer::forward / ParametricResidualLayer::forward (Layer_Base.h:64-95,
// Layers.h:347-361), RACER::Train (Learners/RACER_train.cpp:14-67), Continuous_policy
// (Math/Continuous_policy.h:68-378, 569-738), MiniBatch::setMseDklImpw / setValues
// (MiniBatch.h:161-175), Layer::backward (Layers.h:123-160, 363-393).
#include "tail_dev.h"

namespace hl {

#define FLDR 258            // leading dimension of 16-row LDS tiles (== 2 mod 32: conflict-free MFMA operand reads)
#define FLDS 34             // leading dimension of the 16 x dS state tile
#define FMAXK4 8            // dS <= 32: at most 8 MFMA k-steps in the first layer
#ifndef FUSED_NT
#define FUSED_NT 512        // threads per workgroup of the fused kernel
#endif

// development time stamps of workgroup (panel 0, tile 1), 100 MHz clock: -DHL_FSTAMPS (its own flag: the tail stamps of
// -DHL_TAIL_STAMPS use the same slots of DevScalars::dbgT)
#if defined(HL_FSTAMPS)
#ifndef HL_FSTAMP_PANEL
#define HL_FSTAMP_PANEL 0      // (another panel: a workgroup in the crowd of a larger batch)
#endif
#define FSTAMP(i) do { if (threadIdx.x == 0 && panel == HL_FSTAMP_PANEL && n == 1) a.sc->dbgT[i] = wall_clock64(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

// (-DHL_FSTAMPS) when does the LAST workgroup of each kind finish?  dbgT[20] ordinary panels, [21] panels with next-state rows (they
// start with a dependent load of the row count), [22] the sampler rider, [23] the far-policy / beta rider -- maxima of a monotonic clock:
// after a call they belong to its last launch, like the entry stamp dbgT[31]
#if defined(HL_FSTAMPS)
#define FEND(i) do { __syncthreads(); if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&a.sc->dbgT[i]), (unsigned long long)wall_clock64()); } while (0)
#else
#define FEND(i) do { } while (0)
#endif

// development: stop after phase n (tools/ktime4.py; library built with HL_EXTRA_FLAGS=-DHL_DEV) -- compiled out otherwise
#ifdef HL_DEV
#define FVARIANT_STOP(n) do { if (a.variant == (n)) return; } while (0)
#else
#define FVARIANT_STOP(n) do { } while (0)
#endif

// LDS carve-up (floats).  R1: h1 panel, later the W1 row tile of the dX contraction.  R2: W0
// (k-major, leading dimension H+16), later the y3 panel.  R3: W1 column tile, later f'(x2) ->
// delta_x2 panel.
__host__ __device__ inline int fusedR2Floats(int dSp, int H) { const int a = dSp * (H + 16), b = 16 * FLDR; return a > b ? a : b; }
__host__ __device__ inline int fusedR3Floats(int H) { const int a = H * 16, b = 16 * FLDR; return a > b ? a : b; }
__host__ __device__ inline size_t fusedLdsBytes(int dS, int H) {
  const int dSp = (dS + 3) & ~3;
  const size_t fl = (size_t)16 * FLDR + fusedR2Floats(dSp, H) + fusedR3Floats(H) + (size_t)H * 8 /*Wout*/ + 16 * FLDS +
                    2048 /*red*/ + 3 * (size_t)H /*b0, wres, bres*/ + 512 /*sO*/ + 128 /*sDo*/ + 256 /*own tile scratch*/ + 32 /*bo, bp*/ + 4 /*beta hand-off*/;
  const size_t bytes = fl * 4;
  return bytes > TAIL_LDS_BYTES ? bytes : TAIL_LDS_BYTES;
}

// C[16x16] partial of one wave: sum over its k range of A[i][k] * B[k][j]; operands read from LDS
// through the two index functors, ALL of them before the first MFMA (one exposed LDS latency)
template <int NK, class FA, class FB>
__device__ __forceinline__ f32x4 waveMma(FA fa, FB fb) {
  float av[NK], bv[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) { av[s] = fa(s); bv[s] = fb(s); }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc1, 0, 0, 0);
    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc0, 0, 0, 0);
  }
  return acc0 + acc1;
}

// NT threads per workgroup (256 or 512): threads 0..255 own one output element / one (sample, action
// component) each; with 512 the panel-wide phases and the K-split contractions are spread over 8 waves
// sum of the K-split partial tiles (fixed association: pairs, then pairs of pairs)
template <int NP> __device__ __forceinline__ float redSum(const float* red, int tid) {
  if constexpr (NP == 8) return ((red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid])) + ((red[1024 + tid] + red[1280 + tid]) + (red[1536 + tid] + red[1792 + tid]));
  else return (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}

template <int H, int CF, int NT>
__global__ __launch_bounds__(NT, NT / 128) void fused_fwd_head_dx_kernel(FusedArgs a, ExtraArgs extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // blocks 0..7: riders (tail work of the neighbouring steps); 8 of them keep blockIdx % 8 == XCD
  if (blockIdx.x < 8) {
    if (threadIdx.x >= 256) return;            // the tail code is written for 256 threads
    // the sampler runs in block 0; with PH_PUBLISH its gather is spread over blocks 1..7
    if (blockIdx.x == 0) { if (extra.role == 1) { samplePhases(extra.samp, extra.phases, smem); FEND(22); } }      // (the bookkeeping never rides here: its register needs exceed this kernel's 128)
    else if (blockIdx.x == 1 && a.deferBeta) { farBetaPhase(extra.post, smem); if (threadIdx.x == 0) a.sc->dbgT[23] = wall_clock64(); }      // what the bookkeeping of the step before left over
    else if (extra.role == 1 && (extra.phases & PH_PUBLISH)) gatherHelper(extra.samp, blockIdx.x - 1, 7, smem);
    return;
  }
  constexpr int HT = H / 16, H4 = H / 4, LDW0 = H + 16;
  constexpr int NW = NT / 64;                    // waves per workgroup
  constexpr int KWAVES = (H / 4 >= NW) ? NW : 4; // waves sharing the reduction of the K-split contractions
  constexpr int KW = H / KWAVES;                 // reduction length per wave
  constexpr int NK = KW / 4;                     // MFMA steps per wave
  constexpr int TPW = HT >= NW ? HT / NW : 1;    // h1 column tiles per wave
  constexpr int QP = (16 * H4 + NT - 1) / NT;    // float4 per thread of a 16 x H panel
  constexpr int QC = (H * 4 + NT - 1) / NT;      // float4 per thread of the H x 16 column tile
  constexpr int QO = (H * 2 + NT - 1) / NT;      // float4 per thread of Wout [H][8]
  constexpr int Q0 = (32 * H4 + NT - 1) / NT;    // float4 per thread of W0 (dS <= 32)
  constexpr int QS = 512 / NT;                   // state-tile elements per thread
#if defined(HL_FSTAMPS)
  if (threadIdx.x == 0 && blockIdx.x == 8 + 8) a.sc->dbgT[31] = wall_clock64();   // (panel 0, tile 1): kernel entry
#endif
#ifdef HL_TAIL_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 8 + 8) a.sc->dbgT[31] = wall_clock64();
#endif
#ifdef HL_STEP_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 8 + 8) a.sc->dbgStep[a.sc->nGradSteps & 63] = wall_clock64();      // tools/step_stamps.py
#endif
  const DevScalars* sc = a.sc;
  const int dS = a.dS, dSp = (dS + 3) & ~3, B = a.B, dA = a.dA, nDense = a.nDense;
  // arguments used inside the hot loops, pinned in VGPRs: under SGPR pressure the compiler would
  // otherwise re-load them from the kernarg segment at every use (~1 us per epilogue)
  const int func = __builtin_amdgcn_readfirstlane(a.func);
  int resN = a.resN, ldA0 = a.ldA0, ldA1 = a.ldA1;
  asm volatile("" : "+v"(resN), "+v"(ldA0), "+v"(ldA1));
  float* gR2 = a.R2; float* gX2 = a.X2; float* gD2 = a.D2; float* gDres2 = a.Dres2;
  asm volatile("" : "+v"(gR2), "+v"(gX2), "+v"(gD2), "+v"(gDres2));
  const int bid = blockIdx.x - 8, xcd = bid & 7, gi = bid >> 3;
  const int panel = (gi / HT) * 8 + xcd, n = gi % HT;
  const int m0 = panel * 16, n0 = n * 16;
  // panels made of sampled rows only never wait for the row count of this minibatch
  int nRows = a.B;
  if (m0 + 16 > a.B) { nRows = sc->nRows[a.parity]; if (m0 >= nRows) return; }
  FSTAMP(0);

  float* sY1 = reinterpret_cast<float*>(smem);                 // [16][FLDR]
  float* sR2 = sY1 + 16 * FLDR;
  float* sR3 = sR2 + fusedR2Floats(dSp, H);
  float* sWo = sR3 + fusedR3Floats(H);                         // [H][8]
  float* sS = sWo + H * 8;                                     // [16][FLDS]
  float* red = sS + 16 * FLDS;                                 // [8][256]
  float* sB0 = red + 2048;                                     // [H]
  float* sWr = sB0 + H;                                        // [H] residual w
  float* sBr = sWr + H;                                        // [H] residual b
  double* sO = reinterpret_cast<double*>(sBr + H);             // [16][16]  (offset is a multiple of 8 bytes)
  float* sDo = reinterpret_cast<float*>(sO + 256);             // [16][8]
  float* sT = sDo + 128;                                       // [16][16] own-tile scratch (x1, later delta_y3)
  float* sBo = sT + 256;                                       // [16] output bias, [16] ParamLayer bias
  float* sBp = sBo + 16;
  double* sBeta = reinterpret_cast<double*>(sBp + 16);          // [0] beta of this step, [1] != 0: it has arrived (deferBeta)

  const int tid = threadIdx.x, lane = tid & 63;
  __builtin_assume(tid >= 0 && tid < NT);      // lets the `tid + NT * q < count` guards of full chunks fold away
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches on it
  const int li = lane & 15, lc = lane >> 4;
  const bool eth = tid < 256;                                  // element thread: owns output element / (sample, dim) (em, en)
  const int em = (tid >> 4) & 15, en = tid & 15;
  const float* W = a.W;
  const float* W0 = W + a.indW0; const float* W1 = W + a.indW1; const float* Wo = W + a.indWo;

  // ---- every load that does not depend on the exchange, issued up front --------------------------
  const int row = m0 + em;
  const bool rowValid = eth && row < nRows, isNext = rowValid && row >= B;
  int bSrc = 0; long long slot = 0;
  if (rowValid) { bSrc = isNext ? a.bt.nextSrc[row - B] : row; slot = a.bt.slot[bSrc]; }   // oldest loads: the gathers hang off them
  int eidv = 0;
  if (rowValid && !isNext) eidv = a.bt.eid[row];
  float sv[QS];
#pragma unroll
  for (int q = 0; q < QS; ++q) {
    const int idx = tid + NT * q, r = idx >> 5, c = idx & 31;
    sv[q] = (c < dS && m0 + r < nRows) ? a.X0[(size_t)(m0 + r) * a.ldX0 + c] : 0.f;
  }
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 w0v[Q0], w1c[QC], w1r[QP], wov[QO];
  const int nW0 = dSp * H4;
#pragma unroll
  for (int q = 0; q < Q0; ++q) {
    // no guards: rows >= dS read row dS-1 again (finite weights) and meet state columns that are zero, so they add
    // exactly 0; a guarded load would make the compiler wait for it at the join, ahead of all the loads below
    const int f = tid + NT * q, k = f / H4, c4 = f % H4;
    w0v[q] = *reinterpret_cast<const f32x4*>(W0 + (size_t)(k < dS ? k : dS - 1) * a.ldW0 + 4 * c4);
  }
#pragma unroll
  for (int q = 0; q < QC; ++q) {
    const int f = tid + NT * q; w1c[q] = z4;
    if (f < H * 4) { const int k = f >> 2, c = n0 + (f & 3) * 4; w1c[q] = *reinterpret_cast<const f32x4*>(W1 + (size_t)k * a.ldW1 + c); }
  }
#pragma unroll
  for (int q = 0; q < QO; ++q) { const int f = tid + NT * q; wov[q] = f < H * 2 ? *reinterpret_cast<const f32x4*>(Wo + (size_t)f * 4) : z4; }
  const float b0v = tid < H ? W[a.indB0 + tid] : 0.f;
  const float wrv = tid < H ? W[a.indWr + tid] : 0.f, brv = tid < H ? W[a.indBr + tid] : 0.f;
  const float b1e = eth ? W[a.indB1 + n0 + en] : 0.f;
  const float bov = tid < nDense ? W[a.indBo + tid] : 0.f, bpv = tid < dA ? W[a.indBp + tid] : 0.f;
  double beta = sc->beta; const double Cmax = sc->Cmax, Cinv = sc->Cinv;
  const long long betaWant = sc->nGradSteps;      // (deferBeta: the rider in block 1 publishes beta under this number)
  // ---- stage: states, W0 (k-major), W1 column tile (k-major), Wout, vectors ------------------------
#pragma unroll
  for (int q = 0; q < QS; ++q) { const int idx = tid + NT * q, r = idx >> 5, c = idx & 31; sS[r * FLDS + c] = sv[q]; }
#pragma unroll
  for (int q = 0; q < Q0; ++q) {
    const int f = tid + NT * q;
    if (f < nW0) { const int k = f / H4, c4 = f % H4; *reinterpret_cast<f32x4*>(sR2 + k * LDW0 + 4 * c4) = w0v[q]; }
  }
#pragma unroll
  for (int q = 0; q < QC; ++q) { const int f = tid + NT * q; if (f < H * 4) *reinterpret_cast<f32x4*>(sR3 + (size_t)f * 4) = w1c[q]; }
#pragma unroll
  for (int q = 0; q < QO; ++q) { const int f = tid + NT * q; if (f < H * 2) *reinterpret_cast<f32x4*>(sWo + (size_t)f * 4) = wov[q]; }
  if (tid < H) { sB0[tid] = b0v; sWr[tid] = wrv; sBr[tid] = brv; }
  if (tid < 16) { sBo[tid] = bov; sBp[tid] = bpv; }
  __syncthreads();
  FSTAMP(1);
  FVARIANT_STOP(1);
  // the W1 row tile is needed only by the dX contraction: fetched now, behind the critical first batch
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int f = tid + NT * q; w1r[q] = z4;
    if (f < 16 * H4) { const int r = f / H4, c4 = f % H4; w1r[q] = *reinterpret_cast<const f32x4*>(W1 + (size_t)(n0 + r) * a.ldW1 + 4 * c4); }
  }
  // publishing workgroup of the sample: the aggregates of the sampled episode travel with it to the
  // bookkeeping pass, which then needs no dependent gather (the store happens at the very end)
  // (issued here, behind the first batch and its barrier: they hang off `slot` / `eid`, and a dependent load in the
  // prologue makes everything after it wait for a second HBM round trip)
  const bool aggOwner = ((em & (HT - 1)) == n) && rowValid && !isNext && en < AGG_N;

  // the replay rows of the head (issued now, consumed after the exchange): one (sample, dim) per thread
  double act = 0, bMean = 0, bStd = 1; float misc = 0.f;
  if (rowValid && !isNext && en < dA) {
    act = a.rp.A[(size_t)slot * dA + en];
    bMean = a.rp.MU[(size_t)slot * 2 * dA + en]; bStd = a.rp.MU[(size_t)slot * 2 * dA + dA + en];
  }
  if (rowValid) {   // lanes 0..5: RET, DQ, DKL, IMPW, V, ADV of the sampled step; next rows: lanes 6, 7: V, ADV of t+1
    const float* arr = nullptr; long long sl = slot;
    if (!isNext) arr = en == 0 ? a.rp.RET : en == 1 ? a.rp.DQ : en == 2 ? a.rp.DKL : en == 3 ? a.rp.IMPW : en == 4 ? a.rp.V : en == 5 ? a.rp.ADV : nullptr;
    else { arr = en == 6 ? a.rp.V : en == 7 ? a.rp.ADV : nullptr; sl = slot + 1; }
    if (arr) misc = arr[sl];
  }


  // ---- h1 = f(S W0 + b0), whole panel: wave w computes column tiles w, w+4, ... --------------------
  if (wave < HT) {
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = z4;
    const int nk4 = dSp >> 2;
    // operands of step s+1 are read while the MFMAs of step s run
    float av = sS[li * FLDS + lc], bv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) bv[t] = sR2[lc * LDW0 + (wave + NW * t) * 16 + li];
    for (int s = 0; s < nk4; ++s) {
      const int kn = s + 1 < nk4 ? 4 * (s + 1) + lc : lc;
      const float avn = sS[li * FLDS + kn];
      float bvn[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) bvn[t] = sR2[kn * LDW0 + (wave + NW * t) * 16 + li];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
      av = avn;
#pragma unroll
      for (int t = 0; t < TPW; ++t) bv[t] = bvn[t];
    }
    FSTAMP(14);
    float bb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) bb[t] = sB0[(wave + NW * t) * 16 + li];
    dispatchFunc<CF>(func, [&](auto F) {
      constexpr int FN = decltype(F)::value;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int nt = wave + NW * t;
        const int c = nt * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lc * 4 + r;
          const float x = acc[t][r] + bb[t];
          sY1[i * FLDR + c] = actEvalT<FN>(x);
          if (nt == n) sT[i * 16 + li] = x;
        }
      }
    });
  }
  FSTAMP(15);
  __syncthreads();
  const float x1o = sT[em * 16 + en], y1o = sY1[em * FLDR + n0 + en];
  FSTAMP(2);
  FVARIANT_STOP(2);
  if (eth && row < B) a.Y1[(size_t)row * ldA0 + n0 + en] = y1o;     // A operand of the dW1 contraction
  float aggv = 0.f;
  asm volatile("" : "+v"(eidv));      // keeps the index arithmetic (and the wait for the load) from being hoisted into the prologue
  if (aggOwner) aggv = en < AGG_USED ? a.rp.epAgg[(size_t)eidv * AGG_N + en] : (en == AGG_LEN ? (float)a.rp.epN[eidv] : 0.f);

  // ---- head terms that do not depend on the network outputs of this step (the policy stdev comes
  // from the ParamLayer bias alone).  With 8 waves the element threads (waves 0-3) compute them
  // while waves 4-7 run the x2 contraction; with 4 waves they follow the exchange stores. ------------
  const bool live = rowValid && !isNext;
  const double MAXM = 8.31776613503286;
  double stdev = 1, invStd = 1, dPos = 0, bInv = 1, invVarMu = 1, u2 = 0, lq = 0, CmuCpi = 1;
  bool bnd = false;
  auto headPrecompute = [&]() {
    asm volatile("" : "+v"(act), "+v"(bMean), "+v"(bStd));   // keep the fp64 work (and its wait on the gathers) here
    if (live && en < dA) {
      bnd = ((a.boundedMask >> en) & 1ull) != 0;
      const double pp = (double)sBp[en];
      const double rt = sqrt(1 + pp * pp);
      stdev = (pp + rt) / 2; invStd = 1 / stdev; dPos = (1 + pp / rt) / 2;
      bInv = 1 / bStd; invVarMu = 1 / (bStd * bStd);
      u2 = (act - bMean) * bInv;
      const double qq = stdev * bInv;
      lq = log(qq); CmuCpi = qq * qq;
    }
  };
  constexpr bool SPLITW = (NW == 8 && H >= 16);      // waves 4-7 contract, waves 0-3 do the fp64 terms
  constexpr int XW = SPLITW ? 4 : KWAVES;            // waves sharing the x2 contraction
  constexpr int XKW = H / XW, XNK = XKW / 4;

  // ---- own tile of x2 = h1 W1 + b1: K split over XW waves ------------------------------------------------
  {
    const int xw = SPLITW ? wave - 4 : wave;
    if (xw >= 0 && xw < XW) {
      const int k0 = xw * XKW + lc;
      const f32x4 acc = waveMma<XNK>([&](int s) { return sY1[li * FLDR + k0 + 4 * s]; }, [&](int s) { return sR3[(k0 + 4 * s) * 16 + li]; });
#pragma unroll
      for (int r = 0; r < 4; ++r) red[xw * 256 + (lc * 4 + r) * 16 + li] = acc[r];
    } else if (SPLITW) headPrecompute();
  }
  __syncthreads();
  FSTAMP(3);
  if (rowValid) {
    const float v = redSum<XW>(red, tid);
    const float x2 = v + b1e;
    float y2 = 0.f, f2 = 0.f;
    dispatchFunc<CF>(func, [&](auto F) { constexpr int FN = decltype(F)::value; y2 = actEvalT<FN>(x2); f2 = actDiffT<FN>(x2, y2); });
    const float y3 = (n0 + en < resN) ? resOut(y2, y1o, sWr[n0 + en], sBr[n0 + en]) : y2;
    // plain stores: the consumers are the workgroups of this panel, which share this XCD's L2 (the vector L1 is
    // write-through); other XCDs see y3 after the kernel boundary.  Agent-scope (write-through) stores made the
    // acknowledgement wait below ~1 us longer.
    if (a.xcdSafe) {      // (the probe found workgroups of a panel on different XCDs: through the coherence point)
      __hip_atomic_store(gR2 + (size_t)row * ldA1 + n0 + en, y3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gX2 + (size_t)row * ldA1 + n0 + en, f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      gR2[(size_t)row * ldA1 + n0 + en] = y3;                // also the A operand of dWout
      gX2[(size_t)row * ldA1 + n0 + en] = f2;                // f'(x2)
    }
  }
  FSTAMP(4);
  if (!SPLITW) headPrecompute();
  // ---- group barrier: all HT tiles of this panel are in memory ------------------------------------------
  FSTAMP(5);
  __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): the stores are acknowledged by the L2
  __syncthreads();
  FVARIANT_STOP(3);
  FSTAMP(6);
  // (safe mode: the agent-scope stores above are write-through and acknowledged -- vmcnt(0) -- before the arrival below; the
  // agent-scope loads behind the barrier bypass this XCD's L2: no cache-wide release / acquire, which costs ~20 us each here)
  if (HT > 1 && tid == 0) {
    unsigned* ctr = a.panelCtr + panel * 32;
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned barTarget = (old / (unsigned)HT + 1u) * (unsigned)HT;
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - barTarget) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { a.sc->errFlag = 77; break; }   // never hang the GPU on a lost workgroup
    }
  }
  // beta of this step may still be on its way (POST_DEFER: the count it hangs off is taken by the rider in block 1 while the
  // panels run): one look now, so that the load's latency is hidden behind the output contraction; the wait proper sits in
  // front of the head
  if (a.deferBeta && tid == 0) {
    double got = 0; double ok = 0;
    if (__hip_atomic_load(&a.sc->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == betaWant) {
      got = __hip_atomic_load(&a.sc->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 1;
    }
    sBeta[0] = got; sBeta[1] = ok;
  }
  __syncthreads();

  FSTAMP(7);
  // ---- read the panel's y3 and f'(x2) back -----------------------------------------------------------------
  FVARIANT_STOP(4);
  float* sY3 = sR2; float* sF2 = sR3; float* sBx = sY1;
  f32x4 yv[QP], fv[QP];
  {
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q; yv[q] = z4; fv[q] = z4;
      if (f < 16 * H4) {
        const int r = f / H4, c4 = f % H4;
        if (m0 + r < nRows) {
          if (a.xcdSafe) {
            const float* py = gR2 + (size_t)(m0 + r) * ldA1 + 4 * c4; const float* pf = gX2 + (size_t)(m0 + r) * ldA1 + 4 * c4;
#pragma unroll
            for (int u = 0; u < 4; ++u) { yv[q][u] = __hip_atomic_load(py + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); fv[q][u] = __hip_atomic_load(pf + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          } else {
            yv[q] = *reinterpret_cast<const f32x4*>(gR2 + (size_t)(m0 + r) * ldA1 + 4 * c4);
            fv[q] = *reinterpret_cast<const f32x4*>(gX2 + (size_t)(m0 + r) * ldA1 + 4 * c4);
          }
        }
      }
    }
    // only y3 is needed by the next contraction: f'(x2) and the W1 row tile are staged after it
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q;
      if (f < 16 * H4) {
        const int r = f / H4, c = 4 * (f % H4);
        float2* dy = reinterpret_cast<float2*>(sY3 + r * FLDR + c);
        dy[0] = make_float2(yv[q][0], yv[q][1]); dy[1] = make_float2(yv[q][2], yv[q][3]);
      }
    }
  }
  __syncthreads();
  FSTAMP(8);
  FVARIANT_STOP(5);

  // ---- output layer: O[16][nDense] = y3 Wout + bo (MFMA, columns >= 8 are zero) -----------------------
  if (wave < KWAVES) {
    const int k0 = wave * KW + lc;
    const f32x4 acc = waveMma<NK>([&](int s) { return sY3[li * FLDR + k0 + 4 * s]; },
                                  [&](int s) { return li < 8 ? sWo[(k0 + 4 * s) * 8 + li] : 0.f; });
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
  }
  // h1 is dead (own tile kept in registers): R1 takes the W1 row tile of the dX contraction, R3 f'(x2)
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int f = tid + NT * q;
    if (f < 16 * H4) {
      const int r = f / H4, c = 4 * (f % H4);
      float2* d = reinterpret_cast<float2*>(sBx + r * FLDR + c);
      d[0] = make_float2(w1r[q][0], w1r[q][1]); d[1] = make_float2(w1r[q][2], w1r[q][3]);
      float2* df = reinterpret_cast<float2*>(sF2 + r * FLDR + c);
      df[0] = make_float2(fv[q][0], fv[q][1]); df[1] = make_float2(fv[q][2], fv[q][3]);
    }
  }
  if (a.deferBeta && tid == 0 && sBeta[1] == 0) {
    int spins = 0;
    while (__hip_atomic_load(&a.sc->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != betaWant) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { a.sc->errFlag = 79; break; }
    }
    sBeta[0] = __hip_atomic_load(&a.sc->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (a.deferBeta) beta = sBeta[0];
#ifdef HL_TAIL_STAMPS
  if (tid == 0 && blockIdx.x == 8 + 8) { a.sc->dbgT[13] = wall_clock64(); a.sc->dbgT[3] = sBeta[1] != 0 ? 1 : 0; }      // beta in hand; was it there at the first look
#endif
  FSTAMP(9);
  // network outputs of sample em stay in registers: lane en holds O[em][en] (dense part); the
  // ParamLayer part (Linear) is its bias
  const int base = lane & ~15;
  const float Oen = eth ? redSum<KWAVES>(red, tid) + sBo[en] : 0.f;
  const double O0 = (double)bcast0F(Oen, en);
  const double mean = (double)rowRorF<15>(Oen);                           // O[em][1 + en]: lane i reads lane i+1

  // ---- V-RACER head: thread = (sample em, action component en), fp64 --------------------------------------
  FVARIANT_STOP(6);
  // every workgroup of the group holds the head results of all 16 samples: workgroup n publishes
  // the samples em with em % HT == n, so no single workgroup carries all the stores
  const bool writer = ((em & (HT - 1)) == n);
  if (eth) {
    float g0f = 0.f, gMf = 0.f;
    if (rowValid && isNext) {     // RACER_train.cpp:23-27: V(s_{t+1}) of a truncated episode end
      const float oV = rowRorF<10>(misc), oA = rowRorF<9>(misc);      // lane 0 reads lanes 6 and 7
      if (writer && en == 0) {
        const float Vn = (float)scaleNet2V(O0);
        a.bt.oldNextV[bSrc] = oV; a.bt.oldNextADV[bSrc] = oA;
        a.rp.V[slot + 1] = Vn; a.rp.ADV[slot + 1] = 0.f; a.bt.nextV[bSrc] = Vn;
        a.bt.O[(size_t)row * a.nOut] = O0;
      }
    }
    double lw = 0, kl = 0;
    if (live && en < dA) {
      // log pi(a) - log mu(a) and D_KL(pi || mu) share one logarithm (see head.hip)
      const double m = bnd ? (mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean)) : mean;
      const double u1 = (act - m) * invStd;
      lw = (u2 * u2 - u1 * u1) / 2 - lq;
      const double dm = (mean - bMean) * bInv;
      kl = (CmuCpi - 1 + dm * dm - 2 * lq) / 2;
    }
    const double logW = sum16(lw), DKL = sum16(kl);
    const double RHO = exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
    const float Wf = (float)RHO, Cf = (float)Cmax, iCf = (float)Cinv;
    const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);          // Episode.h:28-33 (Fval)
    const double V = scaleNet2V(O0);
    const double Qret = (double)bcast0F(misc, en);
    const double A_RET = Qret - V, dQ = A_RET;                       // Zero_advantage
    const double Ver = fmin(1.0, RHO) * dQ;
    const double g0 = far ? 0.0 : Ver * beta * scaleVdiff(O0);
    const double coef = A_RET * fmin(Cmax, RHO);
    if (live && en < dA) {
      const double dMean = mean - bMean;
      const double penalM = -1 * (dMean * invVarMu);
      const double penalS = dPos * -1 * ((invVarMu - invStd * invStd) * stdev);
      double polM = 0, polS = 0;
      if (!far) {
        if (bnd) {
          const double dLogPdMean = (act - mean) * invStd * invStd;
          const double m = mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean);
          const double u = (act - m) * invStd;
          polS = dPos * coef * ((u * u - 1) * invStd);
          if (mean >= MAXM && coef * dLogPdMean > 0) polM = 0;
          else if (mean <= -MAXM && coef * dLogPdMean < 0) polM = 0;
          else polM = coef * dLogPdMean;
        } else {
          const double u = (act - mean) * invStd;
          polM = coef * (u * invStd);
          polS = dPos * coef * ((u * u - 1) * invStd);
        }
      }
      const double gM = beta * polM + (1 - beta) * penalM;
      const double gS = beta * polS + (1 - beta) * penalS;
      gMf = (float)gM;                                             // Activation::addOutputDelta: nnReal += Real
      if (writer) {
        a.bt.gParam[(size_t)bSrc * dA + en] = (float)gS;
        a.bt.G[(size_t)bSrc * a.nOut + 1 + en] = (double)gMf;
        a.bt.G[(size_t)bSrc * a.nOut + nDense + en] = (double)(float)gS;
      }
    }
    if (live) g0f = (float)g0;
    // output-layer deltas of the panel (zero for next / padding rows)
    const float gPrev = rowRorF<1>(gMf);                               // component en-1 of the same sample
    if (en < 8) sDo[em * 8 + en] = en == 0 ? g0f : (en <= dA ? gPrev : 0.f);
    if (live && writer) {
      // lane 0 (the one that stores them) reads lanes 1..5
      const float oDQ = rowRorF<15>(misc), oDKL = rowRorF<14>(misc), oW = rowRorF<13>(misc);
      const float oV = rowRorF<12>(misc), oADV = rowRorF<11>(misc);
      if (en == 0) {
        a.bt.pEid[bSrc] = a.bt.eid[bSrc]; a.bt.pNextOf[bSrc] = a.bt.nextOf[bSrc];
        a.bt.G[(size_t)bSrc * a.nOut] = (double)g0f;
        a.bt.rho[bSrc] = RHO; a.bt.dkl[bSrc] = DKL; a.bt.far[bSrc] = far ? 1 : 0;
        // write-backs (Fval casts, MiniBatch.h:161-175); old values kept for the aggregate updates
        const float E = (float)dQ, D = (float)DKL, Wn = (float)RHO, Vf = (float)V;
        a.bt.oldDQ[bSrc] = oDQ; a.bt.oldDKL[bSrc] = oDKL; a.bt.oldW[bSrc] = oW; a.bt.oldV[bSrc] = oV; a.bt.oldADV[bSrc] = oADV;
        a.bt.newDQ[bSrc] = E; a.bt.newDKL[bSrc] = D; a.bt.newW[bSrc] = Wn; a.bt.newV[bSrc] = Vf;
        a.rp.DQ[slot] = E; a.rp.DKL[slot] = D; a.rp.IMPW[slot] = Wn; a.rp.V[slot] = Vf; a.rp.ADV[slot] = 0.f;
        a.bt.dq[bSrc] = (double)E;
      }
      if (en < nDense) a.bt.O[(size_t)row * a.nOut + en] = (double)Oen;
      if (en < dA) a.bt.O[(size_t)row * a.nOut + nDense + en] = (double)sBp[en];
    }
  }
  __syncthreads();
  FSTAMP(10);
  FVARIANT_STOP(7);
  if (eth && writer && row < B && en < nDense) a.dOut[(size_t)row * a.ldDo + en] = sDo[em * 8 + en];

  // ---- delta_y3 = delta_out Wout^T and delta_x2 = delta_y3 f'(x2) are formed directly as the A operand
  // of the dX contraction (8 fused multiply-adds per element; no panel-wide pass, no extra barrier);
  // the element threads compute the same expression for their own element, which goes to global memory ----
  auto deltaY3 = [&](int rowL, int c) {      // sum_o delta_out[rowL][o] * Wout[c][o], fixed association
    const f32x4 wa = *reinterpret_cast<const f32x4*>(sWo + c * 8), wb = *reinterpret_cast<const f32x4*>(sWo + c * 8 + 4);
    const f32x4 da = *reinterpret_cast<const f32x4*>(sDo + rowL * 8), db = *reinterpret_cast<const f32x4*>(sDo + rowL * 8 + 4);
    float s = wa[0] * da[0];
    s = fmaf(wa[1], da[1], s); s = fmaf(wa[2], da[2], s); s = fmaf(wa[3], da[3], s);
    s = fmaf(wb[0], db[0], s); s = fmaf(wb[1], db[1], s); s = fmaf(wb[2], db[2], s); s = fmaf(wb[3], db[3], s);
    return s;
  };
  float dy3own = 0.f;
  if (eth && rowValid) {
    const int c = n0 + en;
    dy3own = deltaY3(em, c);
    if (row < B) { gDres2[(size_t)row * ldA1 + c] = dy3own; gD2[(size_t)row * ldA1 + c] = dy3own * sF2[em * FLDR + c]; }
  }
  FSTAMP(11);
  FVARIANT_STOP(8);

  // ---- own tile of delta_h1 = delta_x2 W1^T (+ residual path), delta_x1 = delta_h1 f'(x1) ----------------------
  if (wave < KWAVES) {
    const int k0 = wave * KW + lc;
    const f32x4 acc = waveMma<NK>([&](int s) { const int c = k0 + 4 * s; return deltaY3(li, c) * sF2[li * FLDR + c]; },
                                  [&](int s) { return sBx[li * FLDR + k0 + 4 * s]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
  }
  __syncthreads();
  FSTAMP(12);
  if (((em & (HT - 1)) == n) && rowValid && !isNext && en < AGG_N) a.bt.aggIn[(size_t)bSrc * AGG_N + en] = aggv;
  if (eth && row < B) {
    const float v = redSum<KWAVES>(red, tid);
    float dres = v;
    if (n0 + en < resN) dres += dy3own * sWr[n0 + en];
    a.Dres1[(size_t)row * ldA0 + n0 + en] = dres;
    float f1 = 1.f;
    dispatchFunc<CF>(func, [&](auto F) { f1 = actDiffT<decltype(F)::value>(x1o, y1o); });
    a.D1[(size_t)row * ldA0 + n0 + en] = dres * f1;
  }
  FSTAMP(13);
  FEND(m0 + 16 > a.B ? 21 : 20);
}

template <int H, int CF>
static hipError_t launchFusedT(const FusedArgs& a, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  const int HT = H / 16, panels = (maxRows + 15) / 16, pg = (panels + 7) / 8;
  const size_t lds = fusedLdsBytes(a.dS, H);
  { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(fused_fwd_head_dx_kernel<H, CF, FUSED_NT>), lds); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL((fused_fwd_head_dx_kernel<H, CF, FUSED_NT>), dim3(8 + 8 * HT * pg), dim3(FUSED_NT), lds, s, a, ex);
  return hipGetLastError();
}
// SoftSign (the reference's default for the shipped settings) and Tanh get their own instantiation;
// every other activation takes the generic one (run-time dispatch inside the epilogues)
template <int H>
static hipError_t launchFusedH(const FusedArgs& a, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  if (a.func == HL_FUNC_SOFTSIGN) return launchFusedT<H, HL_FUNC_SOFTSIGN>(a, maxRows, ex, s);
  if (a.func == HL_FUNC_TANH) return launchFusedT<H, HL_FUNC_TANH>(a, maxRows, ex, s);
  return launchFusedT<H, -1>(a, maxRows, ex, s);
}

hipError_t launch_fused(const FusedArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s) {
  ExtraArgs ex{}; if (extra) ex = *extra;
  switch (a.H) {
    case 16: return launchFusedH<16>(a, maxRows, ex, s);
    case 32: return launchFusedH<32>(a, maxRows, ex, s);
    case 64: return launchFusedH<64>(a, maxRows, ex, s);
    case 128: return launchFusedH<128>(a, maxRows, ex, s);
    case 256: return launchFusedH<256>(a, maxRows, ex, s);
    default: return hipErrorInvalidValue;
  }
}

size_t fused_lds_bytes(int dS, int H) { return fusedLdsBytes(dS, H); }
int fused_threads() { return FUSED_NT; }

// The panel exchange of the fused kernel passes y3 / f'(x2) between the workgroups of a panel with plain stores and loads,
// which is only sound while those workgroups share one XCD's L2 -- i.e. while workgroup b runs on XCD b % 8, an observed
// property of the dispatcher, not a promise of the runtime.  hl_create launches this kernel with the fused kernel's geometry
// (blocks, threads, dynamic LDS) and reads back where every workgroup ran: HW_REG_XCC_ID (hwreg 20, bits 3:0).
__global__ void xcc_probe_kernel(int* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char probeSmem[];
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15);
}
hipError_t launch_xcc_probe(int nBlocks, int nThreads, size_t ldsBytes, int* out, hipStream_t s) {
  { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(xcc_probe_kernel), ldsBytes); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(nBlocks), dim3(nThreads), ldsBytes, s, out);
  return hipGetLastError();
}

}  // namespace hl
