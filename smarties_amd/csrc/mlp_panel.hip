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
#include "head_rows.h"

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

  // ---- the first forward layer's weight tile: the longest cold fetch of the launch goes out first ------------------------------
  GemmProblem Pf{}; TileB tbf; bool haveF = false;
  if (pa.nFwd > 0) { Pf = probs[pa.fwdIdx[0]]; if (n < Pf.tilesN) { gemmLoadB(Pf, panel * Pf.tilesN + n, tbf); haveF = true; } }

  // ---- loads that depend on nothing this launch computes, issued before the forward chain: the sample's replay rows
  // (dependent chain next-row map -> slot -> action / behaviour policy / per-step fields) ... ------------------------------------
  const int row = m0 + em;
  const bool rowValid = row < nRows, isNext = rowValid && row >= B, live = rowValid && !isNext;
  int b = 0; long long slot = 0;
  if (rowValid) { b = isNext ? a.bt.nextSrc[row - B] : row; slot = a.bt.slot[b]; }
  HeadRow<NCH> hr;
  hr.load(a, rowValid, isNext, slot, en);
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
  for (int l = 0; l < pa.nFwd; ++l) {
    if (n < Pf.tilesN) gemmTile<GEMM_ROLE_FWD, GEMM_F, true>(Pf, panel * Pf.tilesN + n, smem, sc, hyp, nRows, haveF ? &tbf : nullptr);
    haveF = false;
    if (l + 1 < pa.nFwd) { Pf = probs[pa.fwdIdx[l + 1]]; if (n < Pf.tilesN) { gemmLoadB(Pf, panel * Pf.tilesN + n, tbf); haveF = true; } }
    panelBarrier(ctr, G, scw);
  }
  PSTMP(2);
  // ---- the panel's rows of the last block's output -> LDS: one batch of loads (H <= 512: eight 16-byte loads per thread at most);
  // while they fly, the head terms that do not depend on this step's network outputs (the policy's standard deviation comes from
  // the ParamLayer bias alone; behaviour-policy terms from the replay rows requested at the top) ------------------------------------
  {
    const int H4 = Hp >> 2;                 // thread (em, en): row em of the panel, 16-byte columns en, en + 16, ... (H <= 512: eight at most)
    f32x4 v[8];
    const float* yRow = a.Yin + (size_t)(m0 + em) * a.ldY;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int c4 = en + 16 * u; v[u] = (c4 < H4 && rowValid) ? *reinterpret_cast<const f32x4*>(yRow + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    hr.hoist(a, pa.boundedMask, bpv, live, en);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c4 = en + 16 * u;
      if (c4 < H4) {
        float2* d = reinterpret_cast<float2*>(sY + em * LY + 4 * c4);
        d[0] = make_float2(v[u][0], v[u][1]); d[1] = make_float2(v[u][2], v[u][3]);
      }
    }
  }
  if (en < 8) sMisc[em * 8 + en] = hr.misc;
  if (en == 0) sAct[em] = hr.actMsg;
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

  // ---- head: thread = (sample em, component / option en + 16 j), fp64 (head_rows.h) -----------------------------------------------
  // every workgroup of the group holds the results of all 16 samples; workgroup (em mod min(G, 16)) publishes sample em
  {
    const int GW = G < 16 ? G : 16;
    hr.compute(a, sO + em * LO, sDelta + em * LD, sXo + em * LD, sMisc + em * 8, sTq + em * 64, sTr + em * 64, rowValid, isNext, (em % GW) == n, b, slot, row, en,
               beta, Cmax, Cinv, sAct[em]);
  }
  PSTMP(6);
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
    if (mine) gemmTile<GEMM_ROLE_DX, GEMM_X, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRows, &tb);
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
