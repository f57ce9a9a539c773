// smarties_amd/csrc/mlp_panel.hip -- forward chain + output layer + RACER / V-RACER head + input-gradient chain of a network
// off the fused path as ONE launch (round 4).
//
// The generic step used to be forward chain -> head (one wavefront per sample, VALU dot products) -> dX -> dW: four launches of
// which the head kernel was the longest or second-longest (9.6 - 13 us) in every BASELINE configuration off the fused path.  Here
// the placement of fused.hip carries all of it: a 16-row PANEL of the minibatch belongs to a group of G workgroups on
// blockIdx = const (mod 8) -- one XCD, one L2 --,
//
//   every workgroup:  [forward chain: its column tile of every dense layer (gemm_tile.h), group barrier between layers]
//                     output layer of the whole panel on v_mfma_f32_16x16x4_f32 (Y panel and W_out^T in LDS, K split over the
//                     four wavefronts), the head in fp64 with one (sample, component) per lane of a 16-lane row (components
//                     beyond 16 in further chunks of the same lanes; DPP row rotations for the sums), results published by
//                     workgroup (sample mod G), its column tiles of delta_last = (delta_out W_out^T) f'(x_last) by MFMA,
//                     [group barrier, then its column tile of every input-gradient problem down the stack]
//
// so a dense net of any depth, width <= 512 and head steps in TWO launches like the cfg-NS class (this kernel + the weight
// gradients), recurrent and convolutional nets lose their head launch's 9 - 13 us.  Reference functions: BaseLayer::forward
// (Network/Layers/Layer_Base.h:64-113), ParamLayer (Layers.h:510-546), RACER::Train (Learners/RACER_train.cpp:14-67),
// Continuous_policy (Math/Continuous_policy.h:68-378, 569-810), Gaussian_advantage (Math/Gaus_advantage.h:17-127),
// Discrete_policy / Discrete_advantage (Math/Discrete_policy.h:19-208, Discrete_advantage.h:17-96), MiniBatch::setMseDklImpw /
// setValues (MiniBatch.h:161-175), Layer::backward (Layers.h:123-160).  The arithmetic of the head is head.hip's (the
// one-wavefront-per-sample kernel this replaces and which stays as the fall-back for shapes outside mlp_panel_ok).
#include "gemm_tile.h"

namespace hl {

constexpr int PN_MAXNT = 5;         // 16-column tiles of the output layer (nDense <= 80)

// development time stamps of workgroup (panel 0, member 0), 100 MHz clock: -DHL_PANEL_STAMPS, tools/panel_stamps.py
#ifdef HL_PANEL_STAMPS
#define PSTMP(i) do { if (threadIdx.x == 0 && panel == 0 && n == 0) scw->dbgT[i] = wall_clock64(); } while (0)
#else
#define PSTMP(i) do { } while (0)
#endif

struct PanelGeo {
  int Hp, LY, NT, LW, LD, LO;
  size_t offRed, offO, offXo, offDelta, offMisc, offAct, offBeta, regionA, offWo, total;
};
__host__ __device__ inline PanelGeo panelGeo(int H, int nDense, int nOut, int ldWo, int nAdv) {
  PanelGeo g;
  g.Hp = (H + 15) & ~15;
  g.LY = ((H + 31) & ~31) + 2;            // == 2 (mod 32): conflict-free MFMA operand reads of 16-row tiles
  g.NT = (nDense + 15) / 16;
  g.LW = ldWo;                            // W_out rows [k][ldWo] as in the parameter blob (flat 16-byte copy)
  g.LD = g.NT * 16 + 6;
  g.LO = nOut | 1;
  size_t o = (size_t)16 * g.LY * 4;                                    // sY
  g.offRed = o; { const size_t red = (size_t)4 * g.NT * 256 * 4, tq = nAdv ? (size_t)2 * 16 * 64 * 8 : 0; o += red > tq ? red : tq; }
  g.offO = o; o += (size_t)16 * g.LO * 8;
  g.offXo = o; o += (size_t)16 * g.LD * 4;
  g.offDelta = o; o += (size_t)16 * g.LD * 4;
  g.offMisc = o; o += 16 * 8 * 4;
  g.offAct = o; o += 16 * 8;
  g.offBeta = o; o += 16;
  g.regionA = o;
  if (g.regionA < (size_t)GEMM_LDS) g.regionA = GEMM_LDS;
  if (g.regionA < (size_t)TAIL_LDS_BYTES) g.regionA = TAIL_LDS_BYTES;
  g.regionA = (g.regionA + 15) & ~(size_t)15;
  g.offWo = g.regionA;
  g.total = g.offWo + (size_t)g.Hp * ldWo * 4;
  return g;
}

// group barrier of a panel: monotonic counter, one arrival per workgroup, bounded spin; the stores before it are plain (the
// group shares one XCD's L2, checked by hl_create's probe) and acknowledged (vmcnt(0)) before the arrival
__device__ __forceinline__ void panelBarrier(unsigned* ctr, int G, DevScalars* sc) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / (unsigned)G + 1u) * (unsigned)G;
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { sc->errFlag = 81; break; }      // never hang the GPU on a lost workgroup
    }
  }
  __syncthreads();
}

__device__ __forceinline__ double spD64(double x) { return (x + sqrt(1 + x * x)) / 2; }            // SoftPlus::_eval (Functions.h:541-584)
__device__ __forceinline__ double spDiff64(double x) { return (1 + x / sqrt(1 + x * x)) / 2; }

// output layer of the panel: wave `wave` takes hidden units [wave KW, (wave + 1) KW), NT column tiles of 16 outputs; partial
// tiles -> red[(wave NT + t)][16 x 16].  pA: this lane's row of the Y panel at its first k; pB: W_out row of that k at output
// column min(li, ldWo - 1) -- columns beyond ldWo feed result columns nobody reads
template <int NT>
__device__ __forceinline__ void panelOutMma(const float* pA, const float* sWoK, int ldWo, int li, int steps, float* redW) {
  f32x4 acc[NT];
  const float* pB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; const int o = t * 16 + li; pB[t] = sWoK + (o < ldWo ? o : ldWo - 1); }
  const int stride = 4 * ldWo;
  constexpr int UN = NT <= 2 ? 8 : 4;             // steps whose operands are in flight together (one exposed LDS latency per batch)
  for (int s0 = 0; s0 < steps; s0 += UN) {
    float av[UN], bv[NT][UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int sc_ = s0 + u < steps ? s0 + u : steps - 1;      // (clamped: no predicated loads; the surplus steps multiply by zero)
      av[u] = pA[4 * sc_];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t][u] = pB[t][(size_t)sc_ * stride];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float a_ = s0 + u < steps ? av[u] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, bv[t][u], acc[t], 0, 0, 0);
    }
  }
  const int lane = threadIdx.x & 63, lc = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) redW[t * 256 + (lc * 4 + r) * 16 + li] = acc[t][r];
  }
}

// NCH: chunks of 16 action components / options per sample row (1: <= 16, 2: <= 32)
template <int NCH>
__global__ __launch_bounds__(256) void mlp_panel_kernel(const GemmProblem* __restrict__ probs, PanelArgs pa, const DevScalars* __restrict__ sc,
                                                        AdamHyper hyp, ExtraArgs extra, ExtraArgs extra2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nR = pa.nRiders;
  if ((int)blockIdx.x < nR) {
    const int rb = blockIdx.x;
    if (rb == 0) { if (extra.role) runExtra(extra, smem); }
    else if (rb == 1) { if (extra2.role) runExtra(extra2, smem); }
    else if (extra.role == 1 && rb - 2 < extra.helpers) gatherHelper(extra.samp, rb - 2, extra.helpers, smem);
    return;
  }
  const HeadArgs& a = pa.h;
  const int bid = blockIdx.x - nR, xcd = bid & 7, gi = bid >> 3;
  const int G = pa.G, panel = (gi / G) * 8 + xcd, n = gi % G;
  const int m0 = panel * 16;
  const int B = a.B;
  int nRows = B;
  if (m0 + 16 > B || pa.nFwd > 0) { nRows = sc->nRows[a.parity]; if (m0 >= nRows) return; }      // (the whole group of a panel leaves together)
  DevScalars* scw = const_cast<DevScalars*>(sc);
  unsigned* ctr = pa.panelCtr + panel * 32;
  PSTMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lc = lane >> 4;
  const int em = tid >> 4, en = tid & 15;
  const int dA = a.dA, nDense = a.nDense, H = a.H, nAdv = a.nAdv, nOpt = a.nOpt, nSig = a.nSig, pM = 1 + nAdv, nOut = a.nOut, ldWo = a.ldWo;
  const bool hasAdv = nAdv > 0 || nOpt > 0;
  const PanelGeo g = panelGeo(H, nDense, nOut, ldWo, nAdv);
  float* sY = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + g.offRed);
  double* sTq = reinterpret_cast<double*>(smem + g.offRed);            // Gaussian advantage: per-component terms (red is dead by then)
  double* sTr = sTq + 16 * 64;
  double* sO = reinterpret_cast<double*>(smem + g.offO);
  float* sXo = reinterpret_cast<float*>(smem + g.offXo);
  float* sDelta = reinterpret_cast<float*>(smem + g.offDelta);
  float* sMisc = reinterpret_cast<float*>(smem + g.offMisc);
  double* sAct = reinterpret_cast<double*>(smem + g.offAct);
  double* sBeta = reinterpret_cast<double*>(smem + g.offBeta);
  float* sWoT = reinterpret_cast<float*>(smem + g.offWo);
  const int LY = g.LY, LW = g.LW, LD = g.LD, LO = g.LO, NT = g.NT, Hp = g.Hp;

  // ---- loads that depend on nothing this launch computes, issued before the forward chain: the sample's replay rows
  // (dependent chain next-row map -> slot -> action / behaviour policy / per-step fields) ... ------------------------------------
  const int row = m0 + em;
  const bool rowValid = row < nRows, isNext = rowValid && row >= B, live = rowValid && !isNext;
  int b = 0; long long slot = 0;
  if (rowValid) { b = isNext ? a.bt.nextSrc[row - B] : row; slot = a.bt.slot[b]; }
  double act[NCH], bMean[NCH], bStd[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = en + 16 * j;
    act[j] = 0; bMean[j] = nOpt ? 1.0 : 0.0; bStd[j] = 1;
    if (live) {
      if (nOpt) { if (c < nOpt) bMean[j] = a.rp.MU[(size_t)slot * nOpt + c]; }       // behaviour probability of option c
      else if (c < dA) { act[j] = a.rp.A[(size_t)slot * dA + c]; bMean[j] = a.rp.MU[(size_t)slot * 2 * dA + c]; bStd[j] = a.rp.MU[(size_t)slot * 2 * dA + dA + c]; }
    }
  }
  double actMsg = 0;
  if (live && nOpt && en == 0) actMsg = a.rp.A[slot];                                 // discrete head: the action message (label + 0.1)
  float misc = 0.f;
  if (rowValid) {   // lanes 0..5: RET, DQ, DKL, IMPW, V, ADV of the sampled step; next rows: lanes 6, 7: V, ADV of t+1
    const float* arr = nullptr; long long sl = slot;
    if (!isNext) arr = en == 0 ? a.rp.RET : en == 1 ? a.rp.DQ : en == 2 ? a.rp.DKL : en == 3 ? a.rp.IMPW : en == 4 ? a.rp.V : en == 5 ? a.rp.ADV : nullptr;
    else { arr = en == 6 ? a.rp.V : en == 7 ? a.rp.ADV : nullptr; sl = slot + 1; }
    if (arr) misc = arr[sl];
  }
  float bov[PN_MAXNT];
#pragma unroll
  for (int t = 0; t < PN_MAXNT; ++t) { const int o = t * 16 + en; bov[t] = (t < NT && o < nDense) ? a.params[a.indBo + o] : 0.f; }
  float bpv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; bpv[j] = c < nSig ? a.params[a.indBp + c] : 0.f; }
  double beta = sc->beta; const double Cmax = sc->Cmax, Cinv = sc->Cinv;
  const long long betaWant = sc->nGradSteps;
  // ... and the output layer's weights into LDS, rows [hidden unit][ldWo] as in the parameter blob (flat 16-byte copy, every load
  // of a batch in flight at once): the B operand of the output contraction and of the back-propagation alike.  Behind the
  // region the forward / input-gradient tiles use.
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.params + a.indWo); f32x4* dst = reinterpret_cast<f32x4*>(sWoT);
    const int total4 = (H * ldWo) >> 2, pad4 = (Hp * ldWo) >> 2;
    for (int f0 = 0; f0 < total4; f0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int f = f0 + tid + 256 * u; v[u] = f < total4 ? src[f] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int f = f0 + tid + 256 * u; if (f < total4) dst[f] = v[u]; }
    }
    for (int f = total4 + tid; f < pad4; f += 256) dst[f] = f32x4{0.f, 0.f, 0.f, 0.f};      // hidden units H .. Hp (the contraction runs over Hp)
  }
  PSTMP(1);
  // ---- forward chain: this workgroup's column tile of every dense layer, the group meets between layers -------------------------
  // (the weight tile of layer l + 1 is requested in front of the barrier behind layer l: it depends on nothing layer l computes)
  {
    GemmProblem P{}; TileB tb; bool have = false;
    if (pa.nFwd > 0) P = probs[pa.fwdIdx[0]];
    for (int l = 0; l < pa.nFwd; ++l) {
      if (n < P.tilesN) gemmTile<GEMM_ROLE_FWD, -1, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRows, have ? &tb : nullptr);
      have = false;
      if (l + 1 < pa.nFwd) { P = probs[pa.fwdIdx[l + 1]]; if (n < P.tilesN) { gemmLoadB(P, panel * P.tilesN + n, tb); have = true; } }
      panelBarrier(ctr, G, scw);
    }
  }
  PSTMP(2);
  // ---- the panel's rows of the last block's output -> LDS: one batch of loads (H <= 512: eight 16-byte loads per thread at most);
  // while they fly, the head terms that do not depend on this step's network outputs (the policy's standard deviation comes from
  // the ParamLayer bias alone; behaviour-policy terms from the replay rows requested at the top) ------------------------------------
  const double MAXM = 8.31776613503286;
  double stdev[NCH], invStd[NCH], dPos[NCH], bInv[NCH], invVarMu[NCH], u2[NCH], lq[NCH], CmuCpi[NCH]; bool bnd[NCH], onC[NCH];
  {
    const int H4 = Hp >> 2;                 // thread (em, en): row em of the panel, 16-byte columns en, en + 16, ... (H <= 512: eight at most)
    f32x4 v[8];
    const float* yRow = a.Yin + (size_t)(m0 + em) * a.ldY;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int c4 = en + 16 * u; v[u] = (c4 < H4 && rowValid) ? *reinterpret_cast<const f32x4*>(yRow + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = en + 16 * j; onC[j] = live && !nOpt && c < dA;
      stdev[j] = 1; invStd[j] = 1; dPos[j] = 0; bInv[j] = 1; invVarMu[j] = 1; u2[j] = 0; lq[j] = 0; CmuCpi[j] = 1; bnd[j] = false;
      if (onC[j]) {
        bnd[j] = ((pa.boundedMask >> c) & 1ull) != 0;
        const double pp = (double)bpv[j];
        const double rt = sqrt(1 + pp * pp);
        stdev[j] = (pp + rt) / 2; invStd[j] = 1 / stdev[j]; dPos[j] = (1 + pp / rt) / 2;
        bInv[j] = 1 / bStd[j]; invVarMu[j] = 1 / (bStd[j] * bStd[j]);
        u2[j] = (act[j] - bMean[j]) * bInv[j];
        const double qq = stdev[j] * bInv[j];
        lq[j] = log(qq); CmuCpi[j] = qq * qq;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c4 = en + 16 * u;
      if (c4 < H4) {
        float2* d = reinterpret_cast<float2*>(sY + em * LY + 4 * c4);
        d[0] = make_float2(v[u][0], v[u][1]); d[1] = make_float2(v[u][2], v[u][3]);
      }
    }
  }
  if (en < 8) sMisc[em * 8 + en] = misc;
  if (en == 0) sAct[em] = actMsg;
  // beta of this step may still be on its way (POST_DEFER): first look now, the wait proper sits in front of the head
  if (pa.deferBeta && tid == 0) {
    double got = 0, ok = 0;
    if (__hip_atomic_load(&scw->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == betaWant) { got = __hip_atomic_load(&scw->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 1; }
    sBeta[0] = got; sBeta[1] = ok;
  }
  // this workgroup's column tiles of the last hidden block (back-propagation below): tile n + G i for wavefront i, i + 4, ...; the
  // pre-activations / outputs its epilogue needs are requested now
  const int HT = (H + 15) >> 4;
  const int myTile = n + G * wave;
  float xl[4], yl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = m0 + lc * 4 + r, cc = myTile * 16 + li;
    const bool ok = myTile < HT && rr < B && cc < H;
    xl[r] = ok ? a.Xlast[(size_t)rr * a.ldD + cc] : 0.f; yl[r] = ok ? a.Ylast[(size_t)rr * a.ldD + cc] : 0.f;
  }
  __syncthreads();
  PSTMP(3);

  // ---- output layer: O[16][nDense] = Y W_out + b_out on MFMA, K split over the four wavefronts -----------------------------------
  {
    const int KW = Hp >> 2, k0 = wave * KW + lc;
    const float* pA = sY + li * LY + k0; const float* sWoK = sWoT + (size_t)k0 * ldWo; float* redW = red + wave * NT * 256;
    switch (NT) {
      case 1: panelOutMma<1>(pA, sWoK, ldWo, li, KW >> 2, redW); break;
      case 2: panelOutMma<2>(pA, sWoK, ldWo, li, KW >> 2, redW); break;
      case 3: panelOutMma<3>(pA, sWoK, ldWo, li, KW >> 2, redW); break;
      case 4: panelOutMma<4>(pA, sWoK, ldWo, li, KW >> 2, redW); break;
      default: panelOutMma<5>(pA, sWoK, ldWo, li, KW >> 2, redW); break;
    }
  }
  __syncthreads();
  PSTMP(4);
#pragma unroll
  for (int t = 0; t < PN_MAXNT; ++t) {
    const int o = t * 16 + en;
    if (t < NT && o < nDense) {      // BaseLayer::forward of the output layer: y = f(x), f = settings nnOutputFunc (Approximator.cpp:228)
      const int e = em * 16 + en;
      const float x = ((red[(0 * NT + t) * 256 + e] + red[(1 * NT + t) * 256 + e]) + (red[(2 * NT + t) * 256 + e] + red[(3 * NT + t) * 256 + e])) + bov[t];
      sXo[em * LD + o] = x; sO[em * LO + o] = (double)(a.outFunc == HL_FUNC_LINEAR ? x : actEval(a.outFunc, x));
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; if (c < nSig) sO[em * LO + nDense + c] = (double)bpv[j]; }      // ParamLayer, Linear
  // zero deltas (padding columns and rows without a gradient: next / absent rows)
  for (int i = tid; i < 16 * LD; i += 256) sDelta[i] = 0.f;
  if (pa.deferBeta && tid == 0 && sBeta[1] == 0) {
    int spins = 0;
    while (__hip_atomic_load(&scw->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != betaWant) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { scw->errFlag = 79; break; }
    }
    sBeta[0] = __hip_atomic_load(&scw->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (pa.deferBeta) beta = sBeta[0];
  PSTMP(5);

  // ---- head: thread = (sample em, component / option en + 16 j), fp64 --------------------------------------------------------------
  // every workgroup of the group holds the results of all 16 samples; workgroup (em mod min(G, 16)) publishes sample em
  const int GW = G < 16 ? G : 16;
  const bool writer = (em % GW) == n;
  const double* O = sO + em * LO;
  const double O0 = O[0];
  if (rowValid && isNext) {     // RACER_train.cpp:23-27: V(s_{t+1}) of a truncated episode end
    if (writer && en == 0) {
      const float Vn = (float)scaleNet2V(O0);
      a.bt.oldNextV[b] = sMisc[em * 8 + 6]; a.bt.oldNextADV[b] = sMisc[em * 8 + 7];
      a.rp.V[slot + 1] = Vn; a.rp.ADV[slot + 1] = 0.f; a.bt.nextV[b] = Vn;
      a.bt.O[(size_t)row * nOut] = O0;
    }
  }
  {
    const double V = scaleNet2V(O0);
    const double Qret = (double)sMisc[em * 8 + 0];
    const float Cf = (float)Cmax, iCf = (float)Cinv;
    double xRHO = 1, xDKL = 0, xdQ = 0, xAval = 0, xg0 = 0; bool xfar = false;
    if (nOpt) {
      // ---- discrete actions: Discrete_policy (SoftPlus-normalised probabilities) and Discrete_advantage; outputs
      // [V | A x nOpt | logits x nOpt], option en + 16 j per lane ----
      const int pA = 1, pP = 1 + nOpt;
      const int label = (int)floor(sAct[em]);                                   // ActionInfo::actionMessage2label
      double logit[NCH], advJ[NCH], unnorm[NCH]; bool on[NCH];
      double su = 0;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = en + 16 * j; on[j] = live && c < nOpt;
        logit[j] = on[j] ? O[pP + c] : 0.0; advJ[j] = on[j] ? O[pA + c] : 0.0;
        unnorm[j] = on[j] ? spD64(logit[j]) : 0.0; su += unnorm[j];
      }
      const double norm = fmax(sum16(su), 2.220446049250313e-16);
      double pj[NCH], lr[NCH], sKl = 0, sEa = 0, sPl = 0, sMl = 0, sAl = 0, sTp = 0, tmp[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = en + 16 * j;
        pj[j] = unnorm[j] / norm;
        const double mj = on[j] ? bMean[j] : 1.0;
        lr[j] = on[j] ? log(pj[j] / mj) : 0.0;
        sKl += on[j] ? pj[j] * lr[j] : 0.0; sEa += on[j] ? pj[j] * advJ[j] : 0.0;
        const bool isL = on[j] && c == label;
        sPl += isL ? pj[j] : 0.0; sMl += isL ? mj : 0.0; sAl += isL ? advJ[j] : 0.0;
        tmp[j] = on[j] ? -(1 + lr[j]) / norm : 0.0; sTp += on[j] ? tmp[j] * pj[j] : 0.0;
      }
      const double RHO = sum16(sPl) / sum16(sMl);                                // importanceWeight (Discrete_policy.h:84-91), no clipping
      const double DKL = sum16(sKl);                                             // KLDivergence (:126-130)
      const float Wf = (float)RHO;
      const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);
      const double Aval = sum16(sAl) - sum16(sEa);                               // computeAdvantage (Discrete_advantage.h:64-70)
      const double tp = sum16(sTp);
      const double A_RET = Qret - V, dQ = A_RET - Aval;
      const double g0 = far ? 0.0 : fmin(1.0, RHO) * dQ * beta * scaleVdiff(O0);
      const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);
#pragma unroll
      for (int j = 0; j < NCH; ++j) if (on[j]) {
        const int c = en + 16 * j;
        const double dpos = spDiff64(logit[j]);
        const double penal = (tmp[j] - tp) * dpos;                               // KLDivGradient(mu, -1) (:152-162)
        double pol = 0;
        if (!far) { const double factor = A_RET * fmin(Cmax, RHO); pol = ((c == label ? factor / unnorm[j] : 0.0) - factor / norm) * dpos; }   // policyGradient (:136-144)
        const double gP = beta * pol + (1 - beta) * penal;                       // penalizeReFER + makeNetworkGrad
        const double gA = Qer * ((c == label ? 1.0 : 0.0) - pj[j]);              // Discrete_advantage::grad (:51-58)
        sDelta[em * LD + pP + c] = (float)gP; sDelta[em * LD + pA + c] = (float)gA;
        if (writer) { a.bt.G[(size_t)b * nOut + pP + c] = (double)(float)gP; a.bt.G[(size_t)b * nOut + pA + c] = (double)(float)gA; }
      }
      xRHO = RHO; xDKL = DKL; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
    } else {
      double mean[NCH], pm[NCH]; bool on[NCH];
      double sLw = 0, sKl = 0;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = en + 16 * j; on[j] = onC[j];
        mean[j] = 0; pm[j] = 0;
        if (on[j]) {
          mean[j] = O[pM + c];
          // log pi(a) - log mu(a) and D_KL(pi || mu) share one logarithm (see head.hip)
          pm[j] = bnd[j] ? (mean[j] > MAXM ? MAXM : (mean[j] < -MAXM ? -MAXM : mean[j])) : mean[j];
          const double u1 = (act[j] - pm[j]) * invStd[j];
          sLw += (u2[j] * u2[j] - u1 * u1) / 2 - lq[j];
          const double dm = (mean[j] - bMean[j]) * bInv[j];
          sKl += (CmuCpi[j] - 1 + dm * dm - 2 * lq[j]) / 2;
        }
      }
      const double logW = sum16(sLw), DKL = sum16(sKl);
      const double RHO = exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
      const float Wf = (float)RHO;
      const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);          // Episode.h:28-33 (Fval)
      // Gaussian_advantage::computeAdvantage (Gaus_advantage.h:76-88): A = coef (exp(-1/2 sum (a-m)^2 / L) - ratio), sums and
      // products in the reference's component order (through LDS: every lane of the row walks the components)
      double Aval = 0, advCoef = 0, advOrig = 0, advRatio = 1, p1[NCH], p2[NCH];
      if (nAdv) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = en + 16 * j; p1[j] = 1; p2[j] = 1;
          if (on[j]) {
            p1[j] = spD64(O[2 + c]); p2[j] = spD64(O[2 + dA + c]);
            const double d = act[j] - pm[j], S = stdev[j] * stdev[j];
            sTq[em * 64 + c] = d * d / (act[j] > pm[j] ? p1[j] : p2[j]);
            sTr[em * 64 + c] = sqrt(p1[j] / (p1[j] + S)) / 2 + sqrt(p2[j] / (p2[j] + S)) / 2;
          }
        }
        __builtin_amdgcn_wave_barrier(); __threadfence_block(); __builtin_amdgcn_wave_barrier();      // (a sample's 16 lanes share a wavefront)
        double quad = 0;
        if (live) for (int i = 0; i < dA; ++i) { quad += sTq[em * 64 + i]; advRatio *= sTr[em * 64 + i]; }
        advCoef = spD64(O[1]); advOrig = exp(-quad / 2);
        Aval = advCoef * (advOrig - advRatio);
      }
      const double A_RET = Qret - V, dQ = A_RET - Aval;                // Zero_advantage: A = 0
      const double Ver = fmin(1.0, RHO) * dQ;
      const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);    // RACER_train.cpp:42,56
      const double g0 = far ? 0.0 : Ver * beta * scaleVdiff(O0);
      const double coef = A_RET * fmin(Cmax, RHO);
#pragma unroll
      for (int j = 0; j < NCH; ++j) if (on[j]) {
        const int c = en + 16 * j;
        const double penalM = -1 * ((mean[j] - bMean[j]) * invVarMu[j]);
        const double penalS = dPos[j] * -1 * ((invVarMu[j] - invStd[j] * invStd[j]) * stdev[j]);
        double polM = 0, polS = 0;
        if (!far) {
          if (bnd[j]) {
            const double dLogPdMean = (act[j] - mean[j]) * invStd[j] * invStd[j];
            const double u = (act[j] - pm[j]) * invStd[j];
            polS = dPos[j] * coef * ((u * u - 1) * invStd[j]);
            if (mean[j] >= MAXM && coef * dLogPdMean > 0) polM = 0;
            else if (mean[j] <= -MAXM && coef * dLogPdMean < 0) polM = 0;
            else polM = coef * dLogPdMean;
          } else {
            const double u = (act[j] - mean[j]) * invStd[j];
            polM = coef * (u * invStd[j]);
            polS = dPos[j] * coef * ((u * u - 1) * invStd[j]);
          }
        }
        const double gM = beta * polM + (1 - beta) * penalM;
        const double gS = beta * polS + (1 - beta) * penalS;
        sDelta[em * LD + pM + c] = (float)gM;                              // Activation::addOutputDelta: nnReal += Real (Activation.h:108-117)
        if (writer) {
          a.bt.gParam[(size_t)b * dA + c] = (float)gS;
          a.bt.G[(size_t)b * nOut + pM + c] = (double)(float)gM;
          a.bt.G[(size_t)b * nOut + nDense + c] = (double)(float)gS;
        }
        if (nAdv) {   // Gaussian_advantage::grad (Gaus_advantage.h:91-116) for the two precisions of this component
          const double expect = -advRatio, S = stdev[j] * stdev[j], d = act[j] - pm[j];
          double g1 = act[j] > pm[j] ? advOrig * advCoef * ((d / p1[j]) * (d / p1[j])) / 2 : 0;
          double g2 = act[j] < pm[j] ? advOrig * advCoef * ((d / p2[j]) * (d / p2[j])) / 2 : 0;
          const double F = 2 / (sqrt(p1[j] / (p1[j] + S)) + sqrt(p2[j] / (p2[j] + S)));
          const double q1 = p1[j] + S, q2 = p2[j] + S;
          g1 += F * expect * advCoef * (S / sqrt(p1[j] * (q1 * q1 * q1)) / 4);
          g2 += F * expect * advCoef * (S / sqrt(p2[j] * (q2 * q2 * q2)) / 4);
          g1 *= Qer * spDiff64(O[2 + c]); g2 *= Qer * spDiff64(O[2 + dA + c]);          // grad_matrix (:69-74)
          sDelta[em * LD + 2 + c] = (float)g1; sDelta[em * LD + 2 + dA + c] = (float)g2;
          if (writer) { a.bt.G[(size_t)b * nOut + 2 + c] = (double)(float)g1; a.bt.G[(size_t)b * nOut + 2 + dA + c] = (double)(float)g2; }
        }
      }
      if (nAdv && live && en == 0) {   // coefficient output of the Gaussian advantage
        const double gc = (advOrig - advRatio) * (Qer * spDiff64(O[1]));
        sDelta[em * LD + 1] = (float)gc;
        if (writer) a.bt.G[(size_t)b * nOut + 1] = (double)(float)gc;
      }
      xRHO = RHO; xDKL = DKL; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
    }
    if (live && en == 0) {
      sDelta[em * LD + 0] = (float)xg0;
      if (writer) {
        a.bt.pEid[b] = a.bt.eid[b]; a.bt.pNextOf[b] = a.bt.nextOf[b];      // (the sampler of the next step overwrites eid / nextOf meanwhile)
        a.bt.G[(size_t)b * nOut] = (double)(float)xg0;
        a.bt.rho[b] = xRHO; a.bt.dkl[b] = xDKL; a.bt.far[b] = xfar ? 1 : 0;
        // write-backs (Fval casts, MiniBatch.h:161-175); old values kept for the aggregate updates
        const float E = (float)xdQ, D = (float)xDKL, Wn = (float)xRHO, Vf = (float)V;
        a.bt.oldDQ[b] = sMisc[em * 8 + 1]; a.bt.oldDKL[b] = sMisc[em * 8 + 2]; a.bt.oldW[b] = sMisc[em * 8 + 3]; a.bt.oldV[b] = sMisc[em * 8 + 4]; a.bt.oldADV[b] = sMisc[em * 8 + 5];
        a.bt.newDQ[b] = E; a.bt.newDKL[b] = D; a.bt.newW[b] = Wn; a.bt.newV[b] = Vf;
        const float Qf = (float)(xAval + V);                    // Episode::updateValues_atomic(t, V, Q): advantage = Q - V in Fval
        a.rp.DQ[slot] = E; a.rp.DKL[slot] = D; a.rp.IMPW[slot] = Wn; a.rp.V[slot] = Vf; a.rp.ADV[slot] = hasAdv ? Qf - Vf : 0.f;
        a.bt.newQ[b] = hasAdv ? Qf : Vf;
        a.bt.dq[b] = (double)E;
      }
    }
    if (live && writer) for (int o = en; o < nOut; o += 16) a.bt.O[(size_t)row * nOut + o] = O[o];
  }
  PSTMP(6);
  __builtin_amdgcn_wave_barrier(); __threadfence_block(); __builtin_amdgcn_wave_barrier();
  // deltas of the output layer: BaseLayer::backward, deltas *= f'(x, y) (Layer_Base.h:104-109)
  if (live) {
    for (int o = en; o < nDense; o += 16) {
      float d = sDelta[em * LD + o];
      if (a.outFunc != HL_FUNC_LINEAR) { d *= actDiff(a.outFunc, sXo[em * LD + o], (float)O[o]); sDelta[em * LD + o] = d; }
      if (writer) a.dOut[(size_t)b * a.ldDo + o] = d;
    }
  }
  __syncthreads();
  PSTMP(7);

  // ---- this workgroup's column tiles of delta_last: Dres = delta_out W_out^T, D = Dres f'(x_last) -------------------------------
  for (int tile = myTile, first = 1; tile < HT; tile += 4 * G, first = 0) {
    const int c0 = tile * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // steps of four outputs, two accumulators; outputs beyond nDense meet zero deltas (sDelta rows are zero padded by six floats),
    // W_out columns beyond ldWo are never addressed (clamped)
    const int nk2 = (nDense + 7) >> 3;
    const float* pD = sDelta + li * LD + lc; const float* pW = sWoT + (size_t)(c0 + li) * ldWo;
    for (int s = 0; s < nk2; ++s) {
      const int oa = 8 * s + lc, ob = oa + 4;
      const float a0 = pD[8 * s], a1 = pD[8 * s + 4];
      const float b0 = pW[oa < ldWo ? oa : ldWo - 1], b1 = pW[ob < ldWo ? ob : ldWo - 1];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = m0 + lc * 4 + r, cc = c0 + li;
      if (rr < B && cc < H) {
        float x = xl[r], y = yl[r];
        if (!first) { x = a.Xlast[(size_t)rr * a.ldD + cc]; y = a.Ylast[(size_t)rr * a.ldD + cc]; }
        a.Dres[(size_t)rr * a.ldD + cc] = acc[r];
        a.D[(size_t)rr * a.ldD + cc] = acc[r] * actDiff(a.func, x, y);
      }
    }
  }

  PSTMP(8);
  // ---- input-gradient chain down the stack: the group meets, then every workgroup takes its column tile of each problem -------------
  for (int l = 0; l < pa.nDx; ++l) {
    const GemmProblem P = probs[pa.dxIdx[l]];
    TileB tb; const bool mine = n < P.tilesN;
    if (mine) gemmLoadB(P, panel * P.tilesN + n, tb);          // W rows of this tile: in flight across the barrier
    panelBarrier(ctr, G, scw);
    if (mine) gemmTile<GEMM_ROLE_DX, -1, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRows, &tb);
  }
  PSTMP(9);
}

bool mlp_panel_ok(const HeadArgs& a) {
  if (a.nDense > PN_MAXNT * 16 || a.H > 512 || a.H < 1) return false;
  if ((a.nOpt ? a.nOpt : a.dA) > 32) return false;
  const PanelGeo g = panelGeo(a.H, a.nDense, a.nOut, a.ldWo, a.nAdv);
  return g.total <= 80 * 1024;          // two workgroups per CU
}
size_t mlp_panel_lds_bytes(const HeadArgs& a) { return panelGeo(a.H, a.nDense, a.nOut, a.ldWo, a.nAdv).total; }
int mlp_panel_blocks(const PanelArgs& pa, int maxRows) { const int panels = (maxRows + 15) / 16, pg = (panels + 7) / 8; return pa.nRiders + 8 * pa.G * pg; }

hipError_t launch_mlp_panel(const GemmProblem* dProbs, const PanelArgs& pa, int maxRows, const DevScalars* sc, const AdamHyper& hyp,
                            const ExtraArgs* extra, const ExtraArgs* extra2, hipStream_t s) {
  ExtraArgs ex{}, ex2{}; if (extra) ex = *extra; if (extra2) ex2 = *extra2;
  const size_t lds = mlp_panel_lds_bytes(pa.h);
  const int nBlk = mlp_panel_blocks(pa, maxRows);
  const int comps = pa.h.nOpt ? pa.h.nOpt : pa.h.dA;
  if (comps <= 16) {
    { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(mlp_panel_kernel<1>), lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((mlp_panel_kernel<1>), dim3(nBlk), dim3(256), lds, s, dProbs, pa, sc, hyp, ex, ex2);
  } else {
    { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(mlp_panel_kernel<2>), lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((mlp_panel_kernel<2>), dim3(nBlk), dim3(256), lds, s, dProbs, pa, sc, hyp, ex, ex2);
  }
  return hipGetLastError();
}

}  // namespace hl
