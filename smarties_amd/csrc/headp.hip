// smarties_amd/csrc/headp.hip -- output layer + RACER / V-RACER head + back-propagation into the last hidden block for a PANEL of
// 16 samples per workgroup (head.hip does the same with one sample per wavefront or per workgroup: the latency form of small
// minibatches).  The pieces are fusedw.hip's second half on their own: the panel of the last hidden block's outputs and its W_out
// in LDS, the output layer as MFMA contractions with K split over the four wavefronts (head_rows.h: panelOutMma), the fp64 head with
// one (sample, action component or option) per lane of a 16-lane row (HeadRow), delta_y = delta_out W_out^T for the whole panel by
// MFMA.  Reference functions: as head.hip (Layer_Base.h:64-113, RACER_train.cpp:14-67, Math/*).
// Used where throughput counts -- local batches of 2048 and more (step_exec.h: launchHead) -- for hidden widths that are a multiple
// of 32 up to 512, up to 80 dense outputs and 32 action components / options.
#include "head_rows.h"

namespace hl {

// development time stamps of the first panel's workgroup (-DHL_HEAD_STAMPS), DevScalars::dbgT[0..] (tools/head_stamps.py panel)
#ifdef HL_HEAD_STAMPS
#define HPSTAMP(i) do { if (m0 == 0 && threadIdx.x == 0) const_cast<DevScalars*>(sc)->dbgT[i] = wall_clock64(); } while (0)
#else
#define HPSTAMP(i) do { } while (0)
#endif
// Two builds: eight wavefronts per workgroup (the shortest chain per panel: up to one panel per CU, 256 panels) and four (the fp64
// head needs 135 registers, which eight wavefronts have only once per CU, four have three times: 49.0 -> 36.8 us at 2048 panels)
constexpr int HP_MAXNT = 5;
struct HpGeo { int LDR, NTo, LD, LO; size_t oF, oWo, oRed, oO, oXo, oDelta, oMisc, oAct, oTq, total; };
__host__ __device__ inline HpGeo hpGeo(int H, int nDense, int nOut, int ldWo, int nAdv, int nWaves) {
  HpGeo g;
  g.LDR = H + 2;                                     // (== 2 mod 32: the 16 rows of a tile fall into different banks)
  g.NTo = (nDense + 15) / 16; g.LD = g.NTo * 16 + 6; g.LO = nOut | 1;
  size_t o = (size_t)16 * g.LDR * 4;                 // sY: the panel's outputs of the last hidden block
  g.oF = o; o += (size_t)16 * g.LDR * 4;             // f'(x) of that block
  g.oWo = o; o += (size_t)H * ldWo * 4;
  g.oRed = o; o += (size_t)nWaves * g.NTo * 256 * 4;
  g.oO = (o + 7) & ~(size_t)7; o = g.oO + (size_t)16 * g.LO * 8;
  g.oXo = o; o += (size_t)16 * g.LD * 4;
  g.oDelta = o; o += (size_t)16 * g.LD * 4;
  g.oMisc = o; o += 16 * 8 * 4;
  g.oAct = (o + 7) & ~(size_t)7; o = g.oAct + 16 * 8;
  g.oTq = (o + 7) & ~(size_t)7; o = g.oTq + (nAdv ? (size_t)2 * 16 * 64 * 8 : 0);      // Gaussian advantage scratch
  g.total = o;
  return g;
}

template <int H, int NCH, int HP_NT>
__global__ __launch_bounds__(HP_NT, HP_NT == 512 ? 2 : 3) void panel_head_kernel(HeadArgs ha, unsigned long long boundedMask, ExtraArgs extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // riders as in head.hip: workgroup 0 (dispatched first) runs sampler phases of the next step, workgroups 1..helpers gather for it
  const int nExtra = extra.role ? 1 + extra.helpers : 0;
  if ((int)blockIdx.x < nExtra) {
    if (threadIdx.x >= 256) return;
    if (blockIdx.x == 0) runExtra(extra, smem); else gatherHelper(extra.samp, blockIdx.x - 1, extra.helpers, smem);
    return;
  }
  constexpr int NT = HP_NT, NW = NT / 64, HT = H / 16, H4 = H / 4;
  constexpr int KW = H / NW, NK = KW / 4;                 // K split of the output layer over the waves (H a multiple of 32)
  constexpr int QP = (16 * H4 + NT - 1) / NT;             // float4 per thread of a 16 x H panel
  const DevScalars* sc = ha.sc;
  const int B = ha.B, nDense = ha.nDense, nOut = ha.nOut, ldWo = ha.ldWo, nSig = ha.nSig;
  const HpGeo g = hpGeo(H, nDense, nOut, ldWo, ha.nAdv, NW);
  const int LDR = g.LDR, NTo = g.NTo, LD = g.LD, LO = g.LO;
  const int m0 = ((int)blockIdx.x - nExtra) * 16;
  int nRows = B;
  if (m0 + 16 > B) { nRows = sc->nRows[ha.parity]; if (m0 >= nRows) return; }
  HPSTAMP(0);
  float* sY = reinterpret_cast<float*>(smem);
  float* sF = reinterpret_cast<float*>(smem + g.oF);
  float* sWo = reinterpret_cast<float*>(smem + g.oWo);               // [H][ldWo]
  float* red = reinterpret_cast<float*>(smem + g.oRed);              // [NW][NTo][256]
  double* sO = reinterpret_cast<double*>(smem + g.oO);
  float* sXo = reinterpret_cast<float*>(smem + g.oXo);
  float* sDelta = reinterpret_cast<float*>(smem + g.oDelta);
  float* sMisc = reinterpret_cast<float*>(smem + g.oMisc);
  double* sAct = reinterpret_cast<double*>(smem + g.oAct);
  double* sTq = reinterpret_cast<double*>(smem + g.oTq); double* sTr = sTq + 16 * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lc = lane >> 4;
  const bool eth = tid < 256;                                        // element thread: owns (em, en)
  const int em = (tid >> 4) & 15, en = tid & 15;
  const float* W = ha.params;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  // ---- every load, up front: the samples' replay rows (dependent chain first), the panel, W_out, biases ---------------------------
  const int row = m0 + em;
  const bool rowValid = eth && row < nRows, isNext = rowValid && row >= B, live = rowValid && !isNext;
  int bSrc = 0; long long slot = 0;
  if (rowValid) { bSrc = isNext ? ha.bt.nextSrc[row - B] : row; slot = ha.bt.slot[bSrc]; }
  HeadRow<NCH> hr;
  hr.load(ha, rowValid, isNext, slot, en);
  // (f'(x) of the last hidden block takes the pre-activation OR the output, never both -- actDiff: one array is read)
  const int func = __builtin_amdgcn_readfirstlane(ha.func);
  const float* Fsrc = actDiffFromOutput(func) ? ha.Ylast : ha.Xlast;
  f32x4 yv[QP], xv[QP];
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int f = tid + NT * q; yv[q] = z4; xv[q] = z4;
    if (f < 16 * H4) {
      const int r = f / H4, c4 = f % H4;
      if (m0 + r < nRows) yv[q] = *reinterpret_cast<const f32x4*>(ha.Yin + (size_t)(m0 + r) * ha.ldY + 4 * c4);
      if (m0 + r < B) xv[q] = *reinterpret_cast<const f32x4*>(Fsrc + (size_t)(m0 + r) * ha.ldD + 4 * c4);
    }
  }
  {      // W_out, rows [hidden unit][ldWo] as in the parameter blob: flat copy
    const f32x4* src = reinterpret_cast<const f32x4*>(W + ha.indWo); f32x4* dst = reinterpret_cast<f32x4*>(sWo);
    const int total4 = (H * ldWo) >> 2;
    for (int f0 = 0; f0 < total4; f0 += NT * 2) {      // (H x ldWo / 4 = 512 sixteen-byte pieces at 256 x 8: one per thread)
      f32x4 v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int f = f0 + tid + NT * u; v[u] = f < total4 ? src[f] : z4; }
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int f = f0 + tid + NT * u; if (f < total4) dst[f] = v[u]; }
    }
  }
  float bov[HP_MAXNT];
#pragma unroll
  for (int t = 0; t < HP_MAXNT; ++t) { const int o = t * 16 + en; bov[t] = (eth && t < NTo && o < nDense) ? W[ha.indBo + o] : 0.f; }
  float bpv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; bpv[j] = (eth && c < nSig) ? W[ha.indBp + c] : 0.f; }
  const double beta = sc->beta, Cmax = sc->Cmax, Cinv = sc->Cinv;
  HPSTAMP(1);
  // ---- stage the panel and f'(x) of the last hidden block ------------------------------------------------------------------------
  dispatchFunc<-1>(func, [&](auto F) {      // (the activation a compile-time constant inside: one switch per workgroup, not one per element)
    constexpr int FN = decltype(F)::value;
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int f = tid + NT * q;
      if (f < 16 * H4) {
        const int r = f / H4, c = 4 * (f % H4);
        float2* dy = reinterpret_cast<float2*>(sY + r * LDR + c);
        dy[0] = make_float2(yv[q][0], yv[q][1]); dy[1] = make_float2(yv[q][2], yv[q][3]);
        f32x4 fp;
#pragma unroll
        for (int e = 0; e < 4; ++e) fp[e] = actDiffT<FN>(xv[q][e], xv[q][e]);
        float2* df = reinterpret_cast<float2*>(sF + r * LDR + c);
        df[0] = make_float2(fp[0], fp[1]); df[1] = make_float2(fp[2], fp[3]);
      }
    }
  });
  if (eth && en < 8) sMisc[em * 8 + en] = hr.misc;
  if (eth && en == 0) sAct[em] = hr.actMsg;
  HPSTAMP(2);
  hr.hoist(ha, boundedMask, bpv, live, en);
  HPSTAMP(3);
  __syncthreads();
  HPSTAMP(4);

  // ---- output layer: O[16][nDense] = y W_out + b_out on MFMA, K split over the waves ------------------------------------------------
  {
    const int k0 = wave * KW + lc;
    const float* pA = sY + li * LDR + k0; const float* sWoK = sWo + (size_t)k0 * ldWo; float* redW = red + wave * NTo * 256;
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
    for (int t = 0; t < HP_MAXNT; ++t) {
      const int o = t * 16 + en;
      if (t < NTo && o < nDense) {      // BaseLayer::forward of the output layer: y = f(x), f = settings nnOutputFunc
        const int e = em * 16 + en;
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 2) x += red[(w * NTo + t) * 256 + e] + red[((w + 1) * NTo + t) * 256 + e];
        x += bov[t];
        sXo[em * LD + o] = x; sO[em * LO + o] = (double)(ha.outFunc == HL_FUNC_LINEAR ? x : actEval(ha.outFunc, x));
      }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) { const int c = en + 16 * j; if (c < nSig) sO[em * LO + nDense + c] = (double)bpv[j]; }      // ParamLayer, Linear
  }
  for (int i = tid; i < 16 * LD; i += NT) sDelta[i] = 0.f;
  __syncthreads();
  HPSTAMP(5);

  // ---- head (head_rows.h): element threads, (sample em, component en + 16 j) -----------------------------------------------------
  if (eth) hr.compute(ha, sO + em * LO, sDelta + em * LD, sXo + em * LD, sMisc + em * 8, sTq + em * 64, sTr + em * 64, rowValid, isNext, true,
                      bSrc, slot, row, en, beta, Cmax, Cinv, sAct[em]);
  HPSTAMP(6);
  __syncthreads();
  HPSTAMP(7);

  // ---- delta_y = delta_out W_out^T for the whole panel by MFMA (wave w: column tiles w, w + 8, ...): gradient w.r.t. the last hidden
  // block's output, and times f'(x) the one w.r.t. its pre-activations (rows of sampled steps only) ----------------------------------
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
      if (rr < B) { const float dy = acc[r]; ha.Dres[(size_t)rr * ha.ldD + c] = dy; ha.D[(size_t)rr * ha.ldD + c] = dy * sF[i * LDR + c]; }
    }
  }
  HPSTAMP(8);
}

template <int H, int NT> static hipError_t panelHeadLaunchT(const HeadArgs& a, unsigned long long mask, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  const size_t lds = std::max(hpGeo(H, a.nDense, a.nOut, a.ldWo, a.nAdv, NT / 64).total, ex.role ? (size_t)TAIL_LDS_BYTES : (size_t)0);      // (the riders' LDS block)
  const int comps = a.nOpt ? a.nOpt : a.dA, nEx = ex.role ? 1 + ex.helpers : 0;
  if (comps <= 16) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(panel_head_kernel<H, 1, NT>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL((panel_head_kernel<H, 1, NT>), dim3((maxRows + 15) / 16 + nEx), dim3(NT), lds, s, a, mask, ex);
  } else {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(panel_head_kernel<H, 2, NT>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL((panel_head_kernel<H, 2, NT>), dim3((maxRows + 15) / 16 + nEx), dim3(NT), lds, s, a, mask, ex);
  }
  return hipGetLastError();
}
template <int H> static hipError_t panelHeadLaunch(const HeadArgs& a, unsigned long long mask, int maxRows, const ExtraArgs& ex, hipStream_t s) {
  // more panels than CUs: the build with three workgroups per CU
  return (maxRows + 15) / 16 > 256 ? panelHeadLaunchT<H, 256>(a, mask, maxRows, ex, s) : panelHeadLaunchT<H, 512>(a, mask, maxRows, ex, s);
}
bool panel_head_ok(const HeadArgs& a) {
  const int H = a.H, comps = a.nOpt ? a.nOpt : a.dA;
  if (!(H == 32 || H == 64 || H == 128 || H == 256 || H == 512) || a.nDense > HP_MAXNT * 16 || comps > 32) return false;
  if ((a.ldY & 3) || (a.ldD & 3) || ((a.H * a.ldWo) & 3) || (a.indWo & 3)) return false;      // 16-byte panel / weight loads
  return hpGeo(H, a.nDense, a.nOut, a.ldWo, a.nAdv, 8).total <= 156 * 1024;      // (two workgroups per CU up to 78 KB: every shape but 512-wide layers)
}
hipError_t launch_panel_head(const HeadArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s) {
  ExtraArgs ex{}; if (extra) ex = *extra;
  unsigned long long mask = 0; for (int c = 0; c < HL_MAX_DIMA && c < 64; ++c) if (a.bounded[c]) mask |= 1ull << c;
  switch (a.H) {
    case 32: return panelHeadLaunch<32>(a, mask, maxRows, ex, s);
    case 64: return panelHeadLaunch<64>(a, mask, maxRows, ex, s);
    case 128: return panelHeadLaunch<128>(a, mask, maxRows, ex, s);
    case 256: return panelHeadLaunch<256>(a, mask, maxRows, ex, s);
    case 512: return panelHeadLaunch<512>(a, mask, maxRows, ex, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace hl
