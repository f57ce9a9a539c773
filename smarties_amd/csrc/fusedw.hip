// smarties_amd/csrc/fusedw.hip -- the fused forward + head + dX kernel (fused.hip) for the two-hidden-layer networks that kernel
// does not take: WIDE states (up to 512 observed components: the first layer is taken tile-wise and exchanged through the panel's L2
// like the second, instead of being recomputed by every workgroup) and ANY head the panel code serves (more than seven action components, the Gaussian and the discrete
// advantage: head_rows.h, one (sample, component) per lane of a 16-lane row, further components in further chunks).
// BASELINE config 3 -- Humanoid through the gym wrapper, 257 states, 17 unbounded actions, 2 x 256, local batch 32 per replica --
// ran the generic launches (forward chain, head, dX, dW: 33.6 us per step); with this kernel it takes the two-kernel step.
// NLH = 3 (end of round 5): a THIRD equal hidden block (settings/RACER_glider.json: 3 x 128, Gaussian advantage: 40.2 -> 27.4 us per step) --
// one more tile stage, panel barrier and read-back each way; its panels and W2 tiles in LDS of their own (fwGeo: oY4 ...).
//
// Placement and exchange are fused.hip's: a 16-row PANEL of the minibatch belongs to the HT = H / 16 workgroups with the same
// blockIdx % 8 (one XCD, one L2); every workgroup takes its 16-column tile of h1 = f(S W0 + b0), panel barrier, reads the panel's
// h1 back, takes its tile of x2 / y3 / f'(x2), second panel barrier, reads the panel back, runs output layer and head for the 16 samples, forms delta_x2 of the whole panel
// locally (delta_y3 = delta_out W_out^T by MFMA, times f'(x2)) and takes its tile of delta_h1 = delta_x2 W1^T.
// Reference functions: as fused.hip (BaseLayer / ParametricResidualLayer forward and backward, RACER::Train, the policies and
// advantages of Math/).
#include "head_rows.h"

namespace hl {

#define WLDR 258            // leading dimension of 16-row LDS tiles (== 2 mod 32)
constexpr int FW_NT = 512;  // threads per workgroup
constexpr int FW_MAXNT = 5; // 16-column tiles of the output layer

// development time stamps of workgroup (panel 0, tile 1), 100 MHz clock: -DHL_PANEL_STAMPS (tools/panel_stamps.py)
#ifdef HL_PANEL_STAMPS
#define WSTMP(i) do { if (threadIdx.x == 0 && panel == 0 && n == 1) scw->dbgT[i] = wall_clock64(); } while (0)
#else
#define WSTMP(i) do { } while (0)
#endif

struct FwGeo { int dSp, LS, NTo, LD, LO; size_t oR2, oR3, oWo, oS, oRed, oVec, oO, oXo, oDelta, oMisc, oAct, oBeta, oT, oY4, oF3, oBx2, oW2c, oVec2, total; };
__host__ __device__ inline FwGeo fwGeo(int dS, int H, int nDense, int nOut, int ldWo, int nAdv, int nLH = 2) {
  FwGeo g;
  g.dSp = (dS + 3) & ~3; g.LS = g.dSp + 2;
  g.NTo = (nDense + 15) / 16; g.LD = g.NTo * 16 + 6; g.LO = nOut | 1;
  size_t o = (size_t)16 * WLDR * 4;                                       // sY1: h1 panel, later the W1 row tile
  g.oR2 = o; { size_t a = (size_t)g.dSp * 16 * 4, b = (size_t)16 * WLDR * 4, c = nAdv ? (size_t)2 * 16 * 64 * 8 : 0; if (b > a) a = b; if (c > a) a = c; o += a; }   // W0 column tile, y3 panel, advantage scratch
  g.oR3 = o; { size_t a = (size_t)H * 16 * 4, b = (size_t)16 * WLDR * 4; o += a > b ? a : b; }     // W1 column tile, later f'(x2) -> delta_x2 panel
  g.oWo = o; o += (size_t)H * ldWo * 4;
  g.oS = o; o += (size_t)16 * g.LS * 4;
  g.oRed = o; o += (size_t)8 * g.NTo * 256 * 4;
  g.oVec = o; o += (size_t)3 * H * 4;
  g.oO = (o + 7) & ~(size_t)7; o = g.oO + (size_t)16 * g.LO * 8;
  g.oXo = o; o += (size_t)16 * g.LD * 4;
  g.oDelta = o; o += (size_t)16 * g.LD * 4;
  g.oMisc = o; o += 16 * 8 * 4;
  g.oAct = (o + 7) & ~(size_t)7; o = g.oAct + 16 * 8;
  g.oBeta = o; o += 16;
  g.oT = o; o += 256 * 4;
  g.oY4 = g.oF3 = g.oBx2 = g.oW2c = g.oVec2 = o;
  if (nLH == 3) {      // a third hidden block: its output panel, f'(x3) -> delta_x3 panel, W2 row tile, W2 column tile, residual vectors
    g.oY4 = o; o += (size_t)16 * WLDR * 4; g.oF3 = o; o += (size_t)16 * WLDR * 4; g.oBx2 = o; o += (size_t)16 * WLDR * 4;
    g.oW2c = o; o += (size_t)H * 16 * 4; g.oVec2 = o; o += (size_t)2 * H * 4;
  }
  g.total = o > (size_t)TAIL_LDS_BYTES ? o : (size_t)TAIL_LDS_BYTES;
  return g;
}

template <int NP> __device__ __forceinline__ float fwRedSum(const float* red, int tid) {
  if constexpr (NP == 8) return ((red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid])) + ((red[1024 + tid] + red[1280 + tid]) + (red[1536 + tid] + red[1792 + tid]));
  else return (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}

// C[16x16] partial of one wave: NK steps, every operand read before the first MFMA
template <int NK, class FA, class FB>
__device__ __forceinline__ f32x4 fwWaveMma(FA fa, FB fb) {
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

// H in {64, 128, 256}; NCH: chunks of 16 action components / options per sample row
template <int H, int NCH, int NLH>
__global__ __launch_bounds__(FW_NT, 2) void fused_wide_kernel(FusedArgs a, HeadArgs ha, ExtraArgs extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x < 8) {      // riders, as in fused.hip: 8 of them keep blockIdx % 8 == XCD for the panels
    if (threadIdx.x >= 256) return;
    if (blockIdx.x == 0) { if (extra.role == 1) samplePhases(extra.samp, extra.phases, smem); }
    else if (blockIdx.x == 1 && a.deferBeta) farBetaPhase(extra.post, smem);
    return;
  }
  constexpr int NT = FW_NT, NW = NT / 64, HT = H / 16, H4 = H / 4;
  constexpr int KW = H / NW, NK = KW / 4;              // K split of the H-long contractions over the 8 waves
  constexpr int QP = (16 * H4 + NT - 1) / NT;          // float4 per thread of a 16 x H panel
  constexpr int QC = (H * 4 + NT - 1) / NT;            // ... of the H x 16 column tile
  const DevScalars* sc = a.sc;
  DevScalars* scw = a.sc;
  const int dS = a.dS, B = a.B, dA = a.dA, nDense = a.nDense, nOut = a.nOut, ldWo = ha.ldWo, nSig = ha.nSig;
  const FwGeo g = fwGeo(dS, H, nDense, nOut, ldWo, ha.nAdv, NLH);
  const int dSp = g.dSp, LS = g.LS, NTo = g.NTo, LD = g.LD, LO = g.LO;
  const int func = __builtin_amdgcn_readfirstlane(a.func);
  int resN = a.resN, ldA0 = a.ldA0, ldA1 = a.ldA1;
  asm volatile("" : "+v"(resN), "+v"(ldA0), "+v"(ldA1));
  float* gR2 = a.R2; float* gX2 = a.X2; float* gD2 = a.D2; float* gDres2 = a.Dres2;
  asm volatile("" : "+v"(gR2), "+v"(gX2), "+v"(gD2), "+v"(gDres2));
  const int bid = blockIdx.x - 8, xcd = bid & 7, gi = bid >> 3;
  const int panel = (gi / HT) * 8 + xcd, n = gi % HT;
  const int m0 = panel * 16, n0 = n * 16;
  int nRows = a.B;
  if (m0 + 16 > a.B) { nRows = sc->nRows[a.parity]; if (m0 >= nRows) return; }

  WSTMP(0);
  float* sY1 = reinterpret_cast<float*>(smem);                       // [16][WLDR]
  float* sR2 = reinterpret_cast<float*>(smem + g.oR2);
  float* sR3 = reinterpret_cast<float*>(smem + g.oR3);
  float* sWo = reinterpret_cast<float*>(smem + g.oWo);               // [H][ldWo]
  float* sS = reinterpret_cast<float*>(smem + g.oS);                 // [16][LS]
  float* red = reinterpret_cast<float*>(smem + g.oRed);              // [8][NTo][256]
  float* sB0 = reinterpret_cast<float*>(smem + g.oVec);
  float* sWr = sB0 + H; float* sBr = sWr + H;
  double* sO = reinterpret_cast<double*>(smem + g.oO);
  float* sXo = reinterpret_cast<float*>(smem + g.oXo);
  float* sDelta = reinterpret_cast<float*>(smem + g.oDelta);
  float* sMisc = reinterpret_cast<float*>(smem + g.oMisc);
  double* sAct = reinterpret_cast<double*>(smem + g.oAct);
  double* sBeta = reinterpret_cast<double*>(smem + g.oBeta);
  float* sT = reinterpret_cast<float*>(smem + g.oT);                 // [16][16] own-tile scratch (x1)
  double* sTq = reinterpret_cast<double*>(sR2);                      // Gaussian advantage scratch (the y3 panel is dead by then)
  double* sTr = sTq + 16 * 64;
  // (NLH == 3) the third block's panels and tiles
  float* sY4 = reinterpret_cast<float*>(smem + g.oY4);               // [16][WLDR] output of the third block (the output layer's input)
  float* sF3 = reinterpret_cast<float*>(smem + g.oF3);               // [16][WLDR] f'(x3) -> delta_x3
  float* sBx2 = reinterpret_cast<float*>(smem + g.oBx2);             // [16][WLDR] W2 row tile
  float* sW2c = reinterpret_cast<float*>(smem + g.oW2c);             // [H][16] W2 column tile
  float* sWr2 = reinterpret_cast<float*>(smem + g.oVec2); float* sBr2 = sWr2 + H;

  const int tid = threadIdx.x, lane = tid & 63;
  __builtin_assume(tid >= 0 && tid < NT);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lc = lane >> 4;
  const bool eth = tid < 256;                                        // element thread: owns (em, en)
  const int em = (tid >> 4) & 15, en = tid & 15;
  const float* W = a.W;
  const float* W0 = W + a.indW0; const float* W1 = W + a.indW1;

  // ---- loads that depend on nothing: the sample's replay rows (dependent chain first), states, first W0 slab, W1 column tile, W_out
  const int row = m0 + em;
  const bool rowValid = eth && row < nRows, isNext = rowValid && row >= B, live = rowValid && !isNext;
  int bSrc = 0; long long slot = 0;
  if (rowValid) { bSrc = isNext ? a.bt.nextSrc[row - B] : row; slot = a.bt.slot[bSrc]; }
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // states: thread (r = tid >> 5, 16-byte columns (tid & 31) + 32 u): dS <= 512 -> four at most
  const int d4 = dSp >> 2;
  f32x4 sv[4];
  {
    const int r = tid >> 5;
    const float* xr = a.X0 + (size_t)(m0 + r) * a.ldX0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c4 = (tid & 31) + 32 * u; sv[u] = (c4 < d4 && m0 + r < nRows) ? *reinterpret_cast<const f32x4*>(xr + 4 * c4) : z4; }
  }
  // this workgroup's 16-column tile of W0 ([dSp][16], rows beyond dS zero): at 257 states recomputing h1 of the whole panel in every
  // workgroup -- fused.hip's way -- is 1040 MFMA steps and a 263 KB weight stream per workgroup (measured: 12.9 us); the tile
  // with an exchange of h1 through the panel's L2 is 65 steps, one more panel barrier and a read-back (3 us)
  constexpr int QW0 = 4;                               // 16-byte loads per thread of the [dSp][16] tile (dS <= 512)
  f32x4 w0v[QW0], w1c[QC], w1r[QP];
#pragma unroll
  for (int q = 0; q < QW0; ++q) {
    const int f = tid + NT * q, k = f >> 2, c = n0 + (f & 3) * 4;
    w0v[q] = (k < dS) ? *reinterpret_cast<const f32x4*>(W0 + (size_t)k * a.ldW0 + c) : z4;
  }
#pragma unroll
  for (int q = 0; q < QC; ++q) {
    const int f = tid + NT * q; w1c[q] = z4;
    if (f < H * 4) { const int k = f >> 2, c = n0 + (f & 3) * 4; w1c[q] = *reinterpret_cast<const f32x4*>(W1 + (size_t)k * a.ldW1 + c); }
  }
  {      // W_out, rows [hidden unit][ldWo] as in the parameter blob: flat copy
    const f32x4* src = reinterpret_cast<const f32x4*>(W + a.indWo); f32x4* dst = reinterpret_cast<f32x4*>(sWo);
    const int total4 = (H * ldWo) >> 2;
    for (int f0 = 0; f0 < total4; f0 += NT * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int f = f0 + tid + NT * u; v[u] = f < total4 ? src[f] : z4; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int f = f0 + tid + NT * u; if (f < total4) dst[f] = v[u]; }
    }
  }
  const float b0v = tid < H ? W[a.indB0 + tid] : 0.f;
  const float wrv = tid < H ? W[a.indWr + tid] : 0.f, brv = tid < H ? W[a.indBr + tid] : 0.f;
  const float b1e = eth ? W[a.indB1 + n0 + en] : 0.f;
  const float* W2 = W + a.indW2;
  f32x4 w2c[QC], w2r[QP];
  float wr2v = 0.f, br2v = 0.f, b2e = 0.f;
  if constexpr (NLH == 3) {
#pragma unroll
    for (int q = 0; q < QC; ++q) {
      const int f = tid + NT * q; w2c[q] = z4;
      if (f < H * 4) { const int k = f >> 2, c = n0 + (f & 3) * 4; w2c[q] = *reinterpret_cast<const f32x4*>(W2 + (size_t)k * a.ldW2 + c); }
    }
    if (tid < H) { wr2v = W[a.indWr2 + tid]; br2v = W[a.indBr2 + tid]; }
    if (eth) b2e = W[a.indB2 + n0 + en];
  }
  float bov[FW_MAXNT];
#pragma unroll
  for (int t = 0; t < FW_MAXNT; ++t) { const int o = t * 16 + en; bov[t] = (eth && t < NTo && o < nDense) ? W[a.indBo + o] : 0.f; }
  float bpv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; bpv[j] = (eth && c < nSig) ? W[a.indBp + c] : 0.f; }
  double beta = sc->beta; const double Cmax = sc->Cmax, Cinv = sc->Cinv;
  const long long betaWant = sc->nGradSteps;
  // ---- stage: states, first slab, W1 column tile, vectors ---------------------------------------------------------------------
  {
    const int r = tid >> 5;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c4 = (tid & 31) + 32 * u;
      if (c4 < d4) { float2* d = reinterpret_cast<float2*>(sS + r * LS + 4 * c4); d[0] = make_float2(sv[u][0], sv[u][1]); d[1] = make_float2(sv[u][2], sv[u][3]); }
    }
  }
#pragma unroll
  for (int q = 0; q < QW0; ++q) { const int f = tid + NT * q; if ((f >> 2) < dSp) *reinterpret_cast<f32x4*>(sR2 + (size_t)f * 4) = w0v[q]; }      // [k][16]
#pragma unroll
  for (int q = 0; q < QC; ++q) { const int f = tid + NT * q; if (f < H * 4) *reinterpret_cast<f32x4*>(sR3 + (size_t)f * 4) = w1c[q]; }
  if (tid < H) { sB0[tid] = b0v; sWr[tid] = wrv; sBr[tid] = brv; }
  if constexpr (NLH == 3) {
#pragma unroll
    for (int q = 0; q < QC; ++q) { const int f = tid + NT * q; if (f < H * 4) *reinterpret_cast<f32x4*>(sW2c + (size_t)f * 4) = w2c[q]; }
    if (tid < H) { sWr2[tid] = wr2v; sBr2[tid] = br2v; }
  }
  __syncthreads();
  WSTMP(1);
  // the W1 row tile is needed only by the dX contraction; the replay rows of the head hang off `slot`
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int f = tid + NT * q; w1r[q] = z4;
    if (f < 16 * H4) { const int r = f / H4, c4 = f % H4; w1r[q] = *reinterpret_cast<const f32x4*>(W1 + (size_t)(n0 + r) * a.ldW1 + 4 * c4); }
  }
  if constexpr (NLH == 3) {
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q; w2r[q] = z4;
      if (f < 16 * H4) { const int r = f / H4, c4 = f % H4; w2r[q] = *reinterpret_cast<const f32x4*>(W2 + (size_t)(n0 + r) * a.ldW2 + 4 * c4); }
    }
  }
  HeadRow<NCH> hr;
  hr.load(ha, rowValid, isNext, slot, en);

  // ---- own tile of h1 = f(S W0 + b0): the dSp / 4 MFMA steps split over the 8 waves (contiguous runs), partial tiles joined in LDS ----
  {
    const int nk4 = dSp >> 2, per = (nk4 + NW - 1) / NW, s0 = wave * per, s1 = min(nk4, s0 + per);
    f32x4 acc0 = z4, acc1 = z4;
    constexpr int UN = 8;
    for (int sb = s0; sb < s1; sb += UN) {
      float av[UN], bv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int sc_ = sb + u < s1 ? sb + u : s1 - 1; av[u] = sS[li * LS + 4 * sc_ + lc]; bv[u] = sR2[(4 * sc_ + lc) * 16 + li]; }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const float a_ = sb + u < s1 ? av[u] : 0.f;
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, bv[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, bv[u], acc0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  }
  __syncthreads();
  WSTMP(2);
  float x1o = 0.f, y1o = 0.f;
  if (rowValid) {
    x1o = fwRedSum<8>(red, tid) + sB0[n0 + en];
    dispatchFunc<-1>(func, [&](auto F) { y1o = actEvalT<decltype(F)::value>(x1o); });
    a.Y1[(size_t)row * ldA0 + n0 + en] = y1o;        // the panel's h1 (exchange below); rows < B: the A operand of the dW1 contraction
  }
  // ---- first panel barrier: all HT tiles of h1 are in memory (plain stores: the group shares one XCD's L2), then the panel is read back ----
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (HT > 1 && tid == 256) {      // (a thread outside the element threads waits for the group: those compute meanwhile)
    unsigned* ctr = a.panelCtr + panel * 32;
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned barTarget = (old / (unsigned)HT + 1u) * (unsigned)HT;
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - barTarget) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { scw->errFlag = 77; break; }
    }
  }
  // head terms that do not depend on this step's outputs: under the barrier's wait
  hr.hoist(ha, a.boundedMask, bpv, live, en);
  __syncthreads();
  {
    f32x4 hv[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q; hv[q] = z4;
      if (f < 16 * H4) { const int r = f / H4, c4 = f % H4; if (m0 + r < nRows) hv[q] = *reinterpret_cast<const f32x4*>(a.Y1 + (size_t)(m0 + r) * ldA0 + 4 * c4); }
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q;
      if (f < 16 * H4) { const int r = f / H4, c = 4 * (f % H4); float2* d = reinterpret_cast<float2*>(sY1 + r * WLDR + c); d[0] = make_float2(hv[q][0], hv[q][1]); d[1] = make_float2(hv[q][2], hv[q][3]); }
    }
  }
  __syncthreads();
  WSTMP(3);

  WSTMP(4);
  // ---- own tile of x2 = h1 W1 + b1: K split over the 8 waves ------------------------------------------------------------------------
  {
    const int k0 = wave * KW + lc;
    const f32x4 acc = fwWaveMma<NK>([&](int s) { return sY1[li * WLDR + k0 + 4 * s]; }, [&](int s) { return sR3[(k0 + 4 * s) * 16 + li]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
  }
  __syncthreads();
  float y3own = 0.f, f2own = 0.f;      // (NLH == 3: the second block's output and f'(x2) of this thread's element stay here)
  if (rowValid) {
    const float v = fwRedSum<8>(red, tid);
    const float x2 = v + b1e;
    float y2 = 0.f, f2 = 0.f;
    dispatchFunc<-1>(func, [&](auto F) { constexpr int FN = decltype(F)::value; y2 = actEvalT<FN>(x2); f2 = actDiffT<FN>(x2, y2); });
    const float y3 = (n0 + en < resN) ? resOut(y2, y1o, sWr[n0 + en], sBr[n0 + en]) : y2;
    gR2[(size_t)row * ldA1 + n0 + en] = y3;                  // plain stores: the consumers share this XCD's L2 (fused.hip); also the A operand of dWout
    if constexpr (NLH == 3) { y3own = y3; f2own = f2; }
    else gX2[(size_t)row * ldA1 + n0 + en] = f2;             // f'(x2)
  }
  WSTMP(5);
  // ---- group barrier: all HT tiles of this panel are in memory ------------------------------------------------------------------------
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (HT > 1 && tid == 0) {
    unsigned* ctr = a.panelCtr + panel * 32;
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned barTarget = (old / (unsigned)HT + 1u) * (unsigned)HT;
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - barTarget) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { scw->errFlag = 77; break; }
    }
  }
  if (a.deferBeta && tid == 0) {
    double got = 0, ok = 0;
    if (__hip_atomic_load(&scw->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == betaWant) { got = __hip_atomic_load(&scw->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 1; }
    sBeta[0] = got; sBeta[1] = ok;
  }
  __syncthreads();

  WSTMP(6);
  // ---- read the panel's y3 and f'(x2) back ---------------------------------------------------------------------------------------
  float* sY3 = sR2; float* sF2 = sR3; float* sBx = sY1;
  {
    f32x4 yv[QP], fv[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q; yv[q] = z4; fv[q] = z4;
      if (f < 16 * H4) {
        const int r = f / H4, c4 = f % H4;
        if (m0 + r < nRows) {
          yv[q] = *reinterpret_cast<const f32x4*>(gR2 + (size_t)(m0 + r) * ldA1 + 4 * c4);
          if constexpr (NLH == 2) fv[q] = *reinterpret_cast<const f32x4*>(gX2 + (size_t)(m0 + r) * ldA1 + 4 * c4);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q;
      if (f < 16 * H4) {
        const int r = f / H4, c = 4 * (f % H4);
        float2* dy = reinterpret_cast<float2*>(sY3 + r * WLDR + c);
        dy[0] = make_float2(yv[q][0], yv[q][1]); dy[1] = make_float2(yv[q][2], yv[q][3]);
        if constexpr (NLH == 2) {
          float2* df = reinterpret_cast<float2*>(sF2 + r * WLDR + c);
          df[0] = make_float2(fv[q][0], fv[q][1]); df[1] = make_float2(fv[q][2], fv[q][3]);
        }
        float2* d = reinterpret_cast<float2*>(sBx + r * WLDR + c);          // h1 is dead (own tile kept in registers): the W1 row tile
        d[0] = make_float2(w1r[q][0], w1r[q][1]); d[1] = make_float2(w1r[q][2], w1r[q][3]);
      }
    }
  }
  if (eth && en < 8) sMisc[em * 8 + en] = hr.misc;
  if (eth && en == 0) sAct[em] = hr.actMsg;
  __syncthreads();

  // ---- (NLH == 3) the third block: own tile of x3 = y3 W2 + b2, its output with the parametric residual of y3, third panel barrier,
  //      read-back of the panel's outputs and f'(x3) -- the same steps as for the second block ----
  float* sYo = sY3; float* sFl = sF2;      // the output layer's input panel / the panel of f'(x) of the last block -> its deltas
  float* gDresL = gDres2; float* gDL = gD2; int ldAL = ldA1;
  if constexpr (NLH == 3) {
    {
      const int k0 = wave * KW + lc;
      const f32x4 acc = fwWaveMma<NK>([&](int s) { return sY3[li * WLDR + k0 + 4 * s]; }, [&](int s) { return sW2c[(k0 + 4 * s) * 16 + li]; });
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
    }
    __syncthreads();
    float* gR3 = a.R3; float* gX3 = a.X3;
    const int ldA2 = a.ldA2;
    if (rowValid) {
      const float x3 = fwRedSum<8>(red, tid) + b2e;
      float y3a = 0.f, f3 = 0.f;
      dispatchFunc<-1>(func, [&](auto F) { constexpr int FN = decltype(F)::value; y3a = actEvalT<FN>(x3); f3 = actDiffT<FN>(x3, y3a); });
      const float r3 = (n0 + en < a.resN2) ? resOut(y3a, y3own, sWr2[n0 + en], sBr2[n0 + en]) : y3a;
      gR3[(size_t)row * ldA2 + n0 + en] = r3;
      gX3[(size_t)row * ldA2 + n0 + en] = f3;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (HT > 1 && tid == 0) {
      unsigned* ctr = a.panelCtr + panel * 32;
      const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned barTarget = (old / (unsigned)HT + 1u) * (unsigned)HT;
      int spins = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - barTarget) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { scw->errFlag = 77; break; }
      }
    }
    __syncthreads();
    {
      f32x4 yv[QP], fv[QP];
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int f = tid + NT * q; yv[q] = z4; fv[q] = z4;
        if (f < 16 * H4) {
          const int r = f / H4, c4 = f % H4;
          if (m0 + r < nRows) {
            yv[q] = *reinterpret_cast<const f32x4*>(gR3 + (size_t)(m0 + r) * ldA2 + 4 * c4);
            fv[q] = *reinterpret_cast<const f32x4*>(gX3 + (size_t)(m0 + r) * ldA2 + 4 * c4);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int f = tid + NT * q;
        if (f < 16 * H4) {
          const int r = f / H4, c = 4 * (f % H4);
          float2* dy = reinterpret_cast<float2*>(sY4 + r * WLDR + c);
          dy[0] = make_float2(yv[q][0], yv[q][1]); dy[1] = make_float2(yv[q][2], yv[q][3]);
          float2* df = reinterpret_cast<float2*>(sF3 + r * WLDR + c);
          df[0] = make_float2(fv[q][0], fv[q][1]); df[1] = make_float2(fv[q][2], fv[q][3]);
          float2* d = reinterpret_cast<float2*>(sBx2 + r * WLDR + c);
          d[0] = make_float2(w2r[q][0], w2r[q][1]); d[1] = make_float2(w2r[q][2], w2r[q][3]);
        }
      }
    }
    __syncthreads();
    sYo = sY4; sFl = sF3; gDresL = a.Dres3; gDL = a.D3; ldAL = ldA2;
  }
  WSTMP(7);
  // ---- output layer: O[16][nDense] = y3 W_out + b_out on MFMA, K split over the 8 waves ---------------------------------------------
  {
    const int k0 = wave * KW + lc;
    const float* pA = sYo + li * WLDR + k0; const float* sWoK = sWo + (size_t)k0 * ldWo; float* redW = red + wave * NTo * 256;
    switch (NTo) {
      case 1: panelOutMma<1>(pA, sWoK, ldWo, li, NK, redW); break;
      case 2: panelOutMma<2>(pA, sWoK, ldWo, li, NK, redW); break;
      case 3: panelOutMma<3>(pA, sWoK, ldWo, li, NK, redW); break;
      case 4: panelOutMma<4>(pA, sWoK, ldWo, li, NK, redW); break;
      default: panelOutMma<5>(pA, sWoK, ldWo, li, NK, redW); break;
    }
  }
  __syncthreads();
  if (eth) {
#pragma unroll
    for (int t = 0; t < FW_MAXNT; ++t) {
      const int o = t * 16 + en;
      if (t < NTo && o < nDense) {      // BaseLayer::forward of the output layer: y = f(x), f = settings nnOutputFunc
        const int e = em * 16 + en;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w += 2) x += red[(w * NTo + t) * 256 + e] + red[((w + 1) * NTo + t) * 256 + e];
        x += bov[t];
        sXo[em * LD + o] = x; sO[em * LO + o] = (double)(ha.outFunc == HL_FUNC_LINEAR ? x : actEval(ha.outFunc, x));
      }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; if (c < nSig) sO[em * LO + nDense + c] = (double)bpv[j]; }      // ParamLayer, Linear
  }
  for (int i = tid; i < 16 * LD; i += NT) sDelta[i] = 0.f;
  if (a.deferBeta && tid == 0 && sBeta[1] == 0) {
    int spins = 0;
    while (__hip_atomic_load(&scw->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != betaWant) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { scw->errFlag = 79; break; }
    }
    sBeta[0] = __hip_atomic_load(&scw->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (a.deferBeta) beta = sBeta[0];
  WSTMP(8);

  // ---- head (head_rows.h): element threads, (sample em, component en + 16 j); workgroup (em mod HT) publishes sample em -----------
  if (eth) hr.compute(ha, sO + em * LO, sDelta + em * LD, sXo + em * LD, sMisc + em * 8, sTq + em * 64, sTr + em * 64, rowValid, isNext, (em & (HT - 1)) == n,
                      bSrc, slot, row, en, beta, Cmax, Cinv, sAct[em]);
  __syncthreads();
  WSTMP(9);

  // ---- delta_y3 = delta_out W_out^T for the whole panel by MFMA (wave w: column tiles w, w + 8, ...), times f'(x2) -> the A operand
  // of the dX contraction, in place of f'(x2); the own tile's delta_y3 / delta_x2 go to memory (B operands of the weight gradients) ----
  float dy3own = 0.f;
  for (int tile = wave; tile < HT; tile += NW) {
    const int c0 = tile * 16;
    f32x4 acc0 = z4, acc1 = z4;
    const int nk2 = (nDense + 7) >> 3;
    const float* pD = sDelta + li * LD + lc; const float* pW = sWo + (size_t)(c0 + li) * ldWo;
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
      const int i = lc * 4 + r, c = c0 + li, rr = m0 + i;
      const float dy3 = acc[r], f2 = sFl[i * WLDR + c];
      sFl[i * WLDR + c] = dy3 * f2;                                       // delta_x of the last block (each element read and written by its own lane)
      if (tile == n && rr < B) { gDresL[(size_t)rr * ldAL + c] = dy3; gDL[(size_t)rr * ldAL + c] = dy3 * f2; sT[i * 16 + li] = dy3; }
    }
  }
  __syncthreads();
  WSTMP(10);
  dy3own = sT[em * 16 + en];
  // ---- (NLH == 3) own tile of the error of the second block's output = delta_x3 W2^T (+ the third block's residual path), its delta_x2;
  //      fourth panel barrier; the panel's delta_x2 read back as the A operand of the last contraction ----
  if constexpr (NLH == 3) {
    {
      const int k0 = wave * KW + lc;
      const f32x4 acc = fwWaveMma<NK>([&](int s) { return sF3[li * WLDR + k0 + 4 * s]; }, [&](int s) { return sBx2[li * WLDR + k0 + 4 * s]; });
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
    }
    __syncthreads();
    float dres2 = 0.f;
    if (eth && row < B) {
      dres2 = fwRedSum<8>(red, tid);
      if (n0 + en < a.resN2) dres2 += dy3own * sWr2[n0 + en];
      gDres2[(size_t)row * ldA1 + n0 + en] = dres2;
      gD2[(size_t)row * ldA1 + n0 + en] = dres2 * f2own;
    }
    dy3own = dres2;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (HT > 1 && tid == 0) {
      unsigned* ctr = a.panelCtr + panel * 32;
      const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned barTarget = (old / (unsigned)HT + 1u) * (unsigned)HT;
      int spins = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - barTarget) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { scw->errFlag = 77; break; }
      }
    }
    __syncthreads();
    {
      f32x4 dv[QP];
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int f = tid + NT * q; dv[q] = z4;
        if (f < 16 * H4) { const int r = f / H4, c4 = f % H4; if (m0 + r < B) dv[q] = *reinterpret_cast<const f32x4*>(gD2 + (size_t)(m0 + r) * ldA1 + 4 * c4); }
      }
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int f = tid + NT * q;
        if (f < 16 * H4) { const int r = f / H4, c = 4 * (f % H4); float2* d = reinterpret_cast<float2*>(sF2 + r * WLDR + c); d[0] = make_float2(dv[q][0], dv[q][1]); d[1] = make_float2(dv[q][2], dv[q][3]); }
      }
    }
    __syncthreads();
  }

  // ---- own tile of delta_h1 = delta_x2 W1^T (+ residual path), delta_x1 = delta_h1 f'(x1) ----------------------------------------------
  {
    const int k0 = wave * KW + lc;
    const f32x4 acc = fwWaveMma<NK>([&](int s) { return sF2[li * WLDR + k0 + 4 * s]; }, [&](int s) { return sBx[li * WLDR + k0 + 4 * s]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc[r];
  }
  __syncthreads();
  if (eth && row < B) {
    const float v = fwRedSum<8>(red, tid);
    float dres = v;
    if (n0 + en < resN) dres += dy3own * sWr[n0 + en];
    a.Dres1[(size_t)row * ldA0 + n0 + en] = dres;
    float f1 = 1.f;
    dispatchFunc<-1>(func, [&](auto F) { f1 = actDiffT<decltype(F)::value>(x1o, y1o); });
    a.D1[(size_t)row * ldA0 + n0 + en] = dres * f1;
  }
  WSTMP(11);
}

template <int H, int NCH, int NLH>
static hipError_t launchFusedWideL(const FusedArgs& a, const HeadArgs& ha, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  const int HT = H / 16, panels = (maxRows + 15) / 16, pg = (panels + 7) / 8;
  const size_t lds = fwGeo(a.dS, H, a.nDense, a.nOut, ha.ldWo, ha.nAdv, NLH).total;
  { hipError_t e = ensureDynLds(reinterpret_cast<const void*>(fused_wide_kernel<H, NCH, NLH>), lds); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL((fused_wide_kernel<H, NCH, NLH>), dim3(8 + 8 * HT * pg), dim3(FW_NT), lds, s, a, ha, ex);
  return hipGetLastError();
}
template <int H, int NCH>
static hipError_t launchFusedWideT(const FusedArgs& a, const HeadArgs& ha, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  return a.nLH == 3 ? launchFusedWideL<H, NCH, 3>(a, ha, maxRows, ex, s) : launchFusedWideL<H, NCH, 2>(a, ha, maxRows, ex, s);
}
template <int H>
static hipError_t launchFusedWideH(const FusedArgs& a, const HeadArgs& ha, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  const int comps = ha.nOpt ? ha.nOpt : ha.dA;
  return comps <= 16 ? launchFusedWideT<H, 1>(a, ha, maxRows, ex, s) : launchFusedWideT<H, 2>(a, ha, maxRows, ex, s);
}
hipError_t launch_fused_wide(const FusedArgs& a, const HeadArgs& ha, int maxRows, const ExtraArgs* extra, hipStream_t s) {
  ExtraArgs ex{}; if (extra) ex = *extra;
  switch (a.H) {
    case 64: return launchFusedWideH<64>(a, ha, maxRows, ex, s);
    case 128: return launchFusedWideH<128>(a, ha, maxRows, ex, s);
    case 256: return launchFusedWideH<256>(a, ha, maxRows, ex, s);
    default: return hipErrorInvalidValue;
  }
}
size_t fused_wide_lds_bytes(int dS, int H, int nDense, int nOut, int ldWo, int nAdv, int nLH) { return fwGeo(dS, H, nDense, nOut, ldWo, nAdv, nLH).total; }
int fused_wide_threads() { return FW_NT; }
bool fused_wide_ok(int dS, int H, int nDense, int nOut, int ldWo, int nAdv, int comps, int nLH) {
  if (!(H == 64 || H == 128 || H == 256) || dS < 1 || dS > 512 || nDense > FW_MAXNT * 16 || comps > 32 || (nLH != 2 && nLH != 3)) return false;
  return fwGeo(dS, H, nDense, nOut, ldWo, nAdv, nLH).total <= 160 * 1024;
}

}  // namespace hl
