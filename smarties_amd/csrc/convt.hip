// smarties_amd/csrc/convt.hip -- the convolutional layers BEHIND the first one with a sample's images and deltas resident in LDS.
//
// Reference: Conv2DLayer<SoftSign, ...>::forward / backward (Network/Layers/Layer_Conv2D.h:88-139), the stacks of
// Network/Builder.cpp:189-203 (84x84x4 -> 8 k8 s4 -> 16 k6 s2 -> 32 k4 s1 -> 64 k3 s1 and 32 k8 s4 -> 64 k4 s2 -> 64 k3 s1).
//
// Behind the first layer a sample's maps are small (RACER_atari.json: 12.8 / 4 / 3.2 KB in, 4 / 3.2 / 2.3 KB out), the per-layer
// launches of conv.hip are dependent chains of 6 - 16 us whose MFMA pipes are busy for 1 - 2 us: each of their (16 positions x 16
// channels) tiles stages 18 - 37 KB of filters for 0.15 MFLOP and gathers its moving operand with 4-byte loads from memory (VERDICT
// r02 - r04: issue stalls 33 - 45 %).  Here ONE workgroup of 16 wavefronts owns a sample and walks the layers with the sample's
// deltas in LDS:
//
//   conv_back_kernel   dL/dX of layers nL-1 .. 1 in one launch (three launches, 37.5 us of the 133 us step until round 4).  Per layer
//                      the SCATTER form of Layer_Conv2D.h:117-138, no zero-padded gather:
//                        T[(ic, fy, fx)][p] = sum_c K[c][ic][fy][fx] D[c][p]              MFMA: M = patch elements, N = output positions,
//                                                                                         reduction over the layer's channels
//                        dIn[ic][iy][ix]    = sum over the taps that reach (iy, ix) of T[(ic, fy, fx)][(iy - fy) / S, (ix - fx) / S]
//                        D_below            = dIn * act'(X_below)
//                      A operand = the filters in the REFERENCE's layout K[c][(ic, fy, fx)] straight from the L2 (a wavefront's load is
//                      four 64-byte runs; every filter element is read once per sample), B operand = the deltas in LDS (padded pitch:
//                      the four channel rows of a step fall into different banks), T goes through LDS, the col2im sum runs in a
//                      fixed order per element (bit-deterministic), the pre-activations are requested in front of the MFMA phase.
//                      832 MFMA steps per sample on the RACER_atari shape where the gather form of conv.hip needs 2096.
#include "dev_common.h"
#include "dw_wide_dev.h"

namespace hl {

constexpr int CT_NT = 1024, CT_NW = CT_NT / 64;
// development time stamps of the first row's workgroup (-DHL_CONVT_STAMPS; 100 MHz clock), DevScalars::dbgT (tools/convt_stamps.py)
#ifdef HL_CONVT_STAMPS
#define CTSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) a.sc->dbgT[i] = wall_clock64(); } while (0)
#else
#define CTSTAMP(i) do { } while (0)
#endif
__device__ __forceinline__ float ctSoftsignDiff(float x) { const float d = 1 + fabsf(x); return 1 / (d * d); }
// LDS pitches: deltas [c][pitch] -- the four channel rows (lc) of an MFMA step 16 banks apart; T [m][pitch] -- the four row groups of
// a result tile 16 banks apart
__host__ __device__ inline int ctPitchD(int P) { const int pp = (P + 15) & ~15; return pp + ((pp & 31) == 0 ? 16 : 0); }
__host__ __device__ inline int ctPitchT(int P) { return ((P + 15) & ~15) + 4; }

// ---- conv_back_kernel -------------------------------------------------------------------------------------------------------------
// A pass = input channels [ic0, ic0 + nIc) of layer l (one pass per layer on the RACER_atari shape; the T buffer bounds nIc).
struct CtPass { int l, ic0, nIc; };
// pl.icPer[l] by constant indices: the plan arrives as scalar kernel arguments, a run-time index would copy it to the stack (scratch)
__device__ __forceinline__ int ctIcPer(const ConvTailPlan& pl, int l) {
  int v = pl.icPer[1];
#pragma unroll
  for (int i = 2; i < HL_MAX_CONV; ++i) v = l == i ? pl.icPer[i] : v;
  return v;
}
__device__ __forceinline__ bool ctNextPass(const ConvGeo* L, const ConvTailPlan& pl, CtPass& p) {      // layers nL-1 .. 1, channels ascending
  const int per = ctIcPer(pl, p.l);
  if (p.ic0 + per < L[p.l].InC) { p.ic0 += per; p.nIc = min(per, L[p.l].InC - p.ic0); return true; }
  if (p.l <= 1) return false;
  --p.l; p.ic0 = 0; p.nIc = min(ctIcPer(pl, p.l), L[p.l].InC);
  return true;
}
// the filter operand of a pass for this wavefront: rows m = (ic, fy, fx) of its two tiles (mt = wave, wave + 16), element c = 4 s + lc of
// the reduction -- K[c][ic0 F + m] in the reference's layout; requested one pass AHEAD (in front of the col2im phase of the pass before)
template <int NST>
__device__ __forceinline__ void ctLoadA(const float* W, const ConvGeo* L, const CtPass& ps, float (&av)[2][16]) {
  const ConvGeo& g = L[ps.l];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lc = lane >> 4;
  const int F = g.KnY * g.KnX, K = g.K, mRows = ps.nIc * F;
  const float* Wl = W + g.indW;                                   // (uniform base, 32-bit element offsets per lane)
  const unsigned o0 = (unsigned)(ps.ic0 * F + lc * K), K4 = 4u * (unsigned)K;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = (wave + t * CT_NW) * 16 + li;
    unsigned o = o0 + (unsigned)(m < mRows ? m : 0);             // (rows beyond the pass: a valid address, their product is never read)
#pragma unroll
    for (int s = 0; s < NST; ++s) { av[t][s] = Wl[o]; o += K4; }
  }
}
// T = K^T D of the pass: a wavefront's (up to) two row tiles x every position tile
template <int NST>
__device__ __forceinline__ void ctBackMfma(const ConvGeo* L, const CtPass& ps, const float (&av)[2][16], const float* __restrict__ sD, float* __restrict__ sT) {
  const ConvGeo& g = L[ps.l];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lc = lane >> 4;
  const int P = g.P, PD = ctPitchD(P), PT = ctPitchT(P), nNt = (P + 15) >> 4;
  const int nMt = (ps.nIc * g.KnY * g.KnX + 15) >> 4;
  const bool two = wave + CT_NW < nMt;
  if (wave >= nMt) return;
  for (int nt = 0; nt < nNt; ++nt) {
    const float* bp = sD + lc * PD + nt * 16 + li;
    float bv[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) bv[s] = bp[4 * s * PD];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[s], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[s], acc0, 0, 0, 0);
    }
    if (two) {
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        if (s & 1) acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[s], acc3, 0, 0, 0);
        else acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[s], acc2, 0, 0, 0);
      }
    }
    float* tp = sT + (wave * 16 + 4 * lc) * PT + nt * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) tp[r * PT] = acc0[r] + acc1[r];
    if (two) {
      float* tp2 = tp + CT_NW * 16 * PT;
#pragma unroll
      for (int r = 0; r < 4; ++r) tp2[r * PT] = acc2[r] + acc3[r];
    }
  }
}
// e / d for 0 <= e < 2^20 and the small divisors of these layers: a multiplication and two corrections instead of the integer division sequence
__device__ __forceinline__ int ctDiv(int e, int d, float inv) {
  int q = (int)(((float)e + 0.5f) * inv);
  if (q * d > e) --q;
  if ((q + 1) * d <= e) ++q;
  return q;
}
// col2im of the pass, act', stores.  KNY > 0: filter and output geometry at compile time (the shapes of Builder.cpp:189-203) -- an
// element's index arithmetic is multiplications by constants, the (KNY / S)(KNX / S) taps that can reach it are LDS reads at
// IMMEDIATE offsets from one base address, all issued before the first addition, and a tap outside the output map is dropped by a
// lane mask (a compare per tap row / column, then one select per tap).  The kernel is bound by its vector instructions, not by the
// LDS or the matrix pipe (16 wavefronts per CU: measured, tools/convt_stamps.py).  KNY = 0: any geometry, loops.
template <int KNY, int KNX, int SS, int OPY, int OPX>
__device__ __forceinline__ void ctBackCol2im(const ConvGeo* L, const CtPass& ps, int b, const float* __restrict__ sT, float* __restrict__ sDn,
                                             float x0, float x1, float x2, float x3) {
  const ConvGeo& g = L[ps.l];
  const ConvGeo& gp = L[ps.l - 1];
  const int tid = threadIdx.x;
  float* Db = gp.D + (size_t)b * gp.ldOut;
  const float* Xb = gp.X + (size_t)b * gp.ldOut;
  if constexpr (KNY > 0) {
    constexpr int TY = KNY / SS, TX = KNX / SS, INY = (OPY - 1) * SS + KNY, INX = (OPX - 1) * SS + KNX, PIN = INY * INX, F = KNY * KNX;
    constexpr int PT = ((OPY * OPX + 15) & ~15) + 4, PDN = ((PIN + 15) & ~15) + ((((PIN + 15) & ~15) & 31) == 0 ? 16 : 0);
    constexpr int SH = SS == 1 ? 0 : (SS == 2 ? 1 : 2);
    static_assert(KNY % SS == 0 && KNX % SS == 0 && (SS == 1 || SS == 2 || SS == 4), "filter sizes are multiples of the stride");
    const unsigned nOut = (unsigned)(ps.nIc * PIN), e0 = (unsigned)(ps.ic0 * PIN);
    auto element = [&](unsigned e, float x) {
      const unsigned icl = e / (unsigned)PIN, q = e - icl * PIN, iy = q / (unsigned)INX, ix = q - iy * INX;
      const unsigned iyc = iy >> SH, ixc = ix >> SH;
      const float* tq = sT + ((icl * F + (iy & (SS - 1)) * KNX + (ix & (SS - 1))) * PT + iyc * OPX + ixc);
      bool okX[TX];
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) okX[tx] = ixc - (unsigned)tx < (unsigned)OPX;
      float sum = 0.f;
#pragma unroll
      for (int ty = 0; ty < TY; ++ty) {      // (fy = py + S ty, oy = iy / S - ty: a tap row's reads in flight together)
        float tv[TX];
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) tv[tx] = tq[ty * (SS * KNX * PT - OPX) + tx * (SS * PT - 1)];
        const bool okY = iyc - (unsigned)ty < (unsigned)OPY;
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) sum += (okY && okX[tx]) ? tv[tx] : 0.f;
      }
      const float d = sum * ctSoftsignDiff(x);
      Db[e0 + e] = d;
      if (sDn) sDn[(ps.ic0 + icl) * PDN + q] = d;
    };
    int it = 0;
    for (unsigned e = tid; e < nOut; e += CT_NT, ++it) {      // (one copy of the element code; the requested pre-activations rotate through x0)
      const float x = it < 4 ? x0 : Xb[e0 + e];
      x0 = x1; x1 = x2; x2 = x3;
      element(e, x);
    }
  } else {
    const int KnX = g.KnX, KnY = g.KnY, Fr = KnY * KnX, PT = ctPitchT(g.P);
    const int InX = g.InX, Pin = g.InY * InX, PDn = ctPitchD(Pin), OpX = g.OpX, OpY = g.OpY;
    const int S = g.S, sh = S == 1 ? 0 : (S == 2 ? 1 : (S == 4 ? 2 : 3));
    const int nOut = ps.nIc * Pin, e0 = ps.ic0 * Pin;
    const float invPin = 1.f / (float)Pin, invInX = 1.f / (float)InX;
    auto element = [&](int e, float x) {
      const int icl = ctDiv(e, Pin, invPin), q = e - icl * Pin, iy = ctDiv(q, InX, invInX), ix = q - iy * InX;
      const float* tq = sT + icl * Fr * PT;
      float sum = 0.f;
      for (int fy = iy & (S - 1); fy < KnY && fy <= iy; fy += S) {
        const int oy = (iy - fy) >> sh;
        if (oy >= OpY) continue;
        for (int fx = ix & (S - 1); fx < KnX && fx <= ix; fx += S) {
          const int ox = (ix - fx) >> sh;
          if (ox < OpX) sum += tq[(fy * KnX + fx) * PT + oy * OpX + ox];
        }
      }
      const float d = sum * ctSoftsignDiff(x);
      Db[e0 + e] = d;
      if (sDn) sDn[(ps.ic0 + icl) * PDn + q] = d;
    };
    int it = 0;
    for (int e = tid; e < nOut; e += CT_NT, ++it) {
      const float x = it < 4 ? x0 : Xb[e0 + e];
      x0 = x1; x1 = x2; x2 = x3;
      element(e, x);
    }
  }
}

__global__ __launch_bounds__(CT_NT) void conv_back_kernel(ConvArgs a, ConvTailPlan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  if (b >= a.B) return;
  CTSTAMP(0);
  const int tid = threadIdx.x, nL = a.nL;
  float* base = reinterpret_cast<float*>(smem);
  float* sT = base + 2 * pl.bufD;
  const int bufD = pl.bufD;
  const float* W = a.W;
  // (the layers' geometry is read from the argument block by scalar loads -- uniform values in scalar registers; a copy in LDS was
  //  measured: the same time, and every geometry value then occupies a vector register)
  const ConvGeo* L = a.L;
  CtPass ps; ps.l = nL - 1; ps.ic0 = 0; ps.nIc = min(ctIcPer(pl, ps.l), L[ps.l].InC);
  {      // the deltas of the last layer (written by the dense layers' input-gradient launch)
    const ConvGeo& g = L[nL - 1];
    const int P = g.P, PD = ctPitchD(P), n = g.KnC * P;
    const float invP = 1.f / (float)P;
    float* dst = base + ((nL - 1) & 1) * bufD;
    const float* src = g.D + (size_t)b * g.ldOut;
    for (int e = tid; e < n; e += CT_NT) { const int c = ctDiv(e, P, invP), p = e - c * P; dst[c * PD + p] = src[e]; }
  }
  __syncthreads();
  CTSTAMP(1);
  int st = 2;
  for (bool more = true; more;) {
    const int l = ps.l;
    const ConvGeo& g = L[l];
    const float* sD = base + (l & 1) * bufD;
    float* sDn = l > 1 ? base + ((l - 1) & 1) * bufD : nullptr;
    // pre-activations of this thread's elements of the layer below (for act'): requested in front of the MFMA phase
    float x0, x1, x2, x3;
    {
      const ConvGeo& gp = L[l - 1];
      const unsigned Pin = (unsigned)(g.InY * g.InX), nOut = (unsigned)ps.nIc * Pin, ut = (unsigned)tid;
      const float* Xb = gp.X + (size_t)b * gp.ldOut + (size_t)ps.ic0 * Pin;
      x0 = Xb[ut < nOut ? ut : 0]; x1 = Xb[ut + CT_NT < nOut ? ut + CT_NT : 0];
      x2 = Xb[ut + 2 * CT_NT < nOut ? ut + 2 * CT_NT : 0]; x3 = Xb[ut + 3 * CT_NT < nOut ? ut + 3 * CT_NT : 0];
    }
    const int nSt = g.KnC >> 2;
    // (the filter operand is requested here, not a pass ahead as in the kernel with the geometry at compile time: carried around the
    //  pass loop the 32 values did not stay in registers)
    if (nSt == 4) { float av[2][16]; ctLoadA<4>(W, L, ps, av); ctBackMfma<4>(L, ps, av, sD, sT); }
    else if (nSt == 8) { float av[2][16]; ctLoadA<8>(W, L, ps, av); ctBackMfma<8>(L, ps, av, sD, sT); }
    else { float av[2][16]; ctLoadA<16>(W, L, ps, av); ctBackMfma<16>(L, ps, av, sD, sT); }
    const CtPass cur = ps;
    more = ctNextPass(L, pl, ps);
    CTSTAMP(st);
    __syncthreads();
    CTSTAMP(st + 1);
    ctBackCol2im<0, 0, 0, 0, 0>(L, cur, b, sT, sDn, x0, x1, x2, x3);
    CTSTAMP(st + 2);
    __syncthreads();
    CTSTAMP(st + 3);
    st += 4;
  }
}


// ---- the same with EVERY layer's geometry at compile time (the stack of RACER_atari.json: 16 k6 s2, 32 k4 s1, 64 k3 s1 behind the
// first layer): pitches, tile counts and tap offsets are constants, the layer loop is unrolled, no geometry value occupies a register.
// The any-geometry kernel above needs ~150 vector instructions per col2im element and keeps ~50 uniform values live; bound by its
// vector instructions (16 wavefronts per CU) it took 19 us where this one is written for the instruction count.
template <int INC_, int KNC_, int KNY_, int KNX_, int SS_, int OPY_, int OPX_>
struct CtShape {
  static constexpr int INC = INC_, KNC = KNC_, KNY = KNY_, KNX = KNX_, SS = SS_, OPY = OPY_, OPX = OPX_;
  static constexpr int F = KNY * KNX, K = INC * F, P = OPY * OPX, INY = (OPY - 1) * SS + KNY, INX = (OPX - 1) * SS + KNX, PIN = INY * INX;
  static constexpr int NST = KNC / 4, NNT = (P + 15) / 16, NMT = (K + 15) / 16, TY = KNY / SS, TX = KNX / SS, SH = SS == 1 ? 0 : (SS == 2 ? 1 : 2);
  static constexpr int PP = (P + 15) & ~15, PD = PP + ((PP & 31) == 0 ? 16 : 0), PT = PP + 4;
  static constexpr int PPN = (PIN + 15) & ~15, PDN = PPN + ((PPN & 31) == 0 ? 16 : 0);
  static_assert(KNC % 4 == 0 && KNY % SS == 0 && KNX % SS == 0 && NMT <= 2 * CT_NW && NST <= 16, "shape");
};
template <class SH>
__device__ __forceinline__ void ctLoadAT(const float* __restrict__ Wl, float (&av)[2][16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lc = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && SH::NMT <= CT_NW) break;
    const int m = (wave + t * CT_NW) * 16 + li;
    const unsigned o = (unsigned)(lc * SH::K + (m < SH::K ? m : 0));
#pragma unroll
    for (int s = 0; s < SH::NST; ++s) av[t][s] = Wl[o + (unsigned)(4 * s * SH::K)];
  }
}
template <class SH>
__device__ __forceinline__ void ctMfmaT(const float (&av)[2][16], const float* __restrict__ sD, float* __restrict__ sT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lc = lane >> 4;
  if (wave >= SH::NMT) return;
  const bool two = SH::NMT > CT_NW && wave + CT_NW < SH::NMT;
  const float* bp = sD + lc * SH::PD + li;
  float* tp = sT + (wave * 16 + 4 * lc) * SH::PT + li;
#pragma unroll
  for (int nt = 0; nt < SH::NNT; ++nt) {
    float bv[SH::NST];
#pragma unroll
    for (int s = 0; s < SH::NST; ++s) bv[s] = bp[4 * s * SH::PD + nt * 16];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SH::NST; ++s) {
      if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[s], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[s], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) tp[r * SH::PT + nt * 16] = acc0[r] + acc1[r];
    if constexpr (SH::NMT > CT_NW) {
      if (two) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SH::NST; ++s) {
          if (s & 1) acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[s], acc3, 0, 0, 0);
          else acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[s], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tp[(CT_NW * 16 + r) * SH::PT + nt * 16] = acc2[r] + acc3[r];
      }
    }
  }
}
template <class SH, bool KEEP>      // KEEP: the layer below has an input gradient of its own to compute (its deltas stay in LDS)
__device__ __forceinline__ void ctCol2imT(const float* __restrict__ sT, float* __restrict__ sDn, float* __restrict__ Db, const float* __restrict__ Xb,
                                          float x0, float x1, float x2, float x3) {
  constexpr unsigned nOut = SH::INC * SH::PIN;
  int it = 0;
  for (unsigned e = threadIdx.x; e < nOut; e += CT_NT, ++it) {
    const float x = it < 4 ? x0 : Xb[e];
    x0 = x1; x1 = x2; x2 = x3;
    const unsigned icl = e / (unsigned)SH::PIN, q = e - icl * SH::PIN, iy = q / (unsigned)SH::INX, ix = q - iy * SH::INX;
    const unsigned iyc = iy >> SH::SH, ixc = ix >> SH::SH;
    const float* tq = sT + ((icl * SH::F + (iy & (SH::SS - 1)) * SH::KNX + (ix & (SH::SS - 1))) * SH::PT + iyc * SH::OPX + ixc);
    bool okX[SH::TX];
#pragma unroll
    for (int tx = 0; tx < SH::TX; ++tx) okX[tx] = ixc - (unsigned)tx < (unsigned)SH::OPX;
    float sum = 0.f;
#pragma unroll
    for (int ty = 0; ty < SH::TY; ++ty) {      // (fy = py + S ty, oy = iy / S - ty: a tap row's reads in flight together)
      float tv[SH::TX];
#pragma unroll
      for (int tx = 0; tx < SH::TX; ++tx) tv[tx] = tq[ty * (SH::SS * SH::KNX * SH::PT - SH::OPX) + tx * (SH::SS * SH::PT - 1)];
      const bool okY = iyc - (unsigned)ty < (unsigned)SH::OPY;
#pragma unroll
      for (int tx = 0; tx < SH::TX; ++tx) sum += (okY && okX[tx]) ? tv[tx] : 0.f;
    }
    const float d = sum * ctSoftsignDiff(x);
    Db[e] = d;
    if (KEEP) sDn[icl * SH::PDN + q] = d;
  }
}
template <class SH>
__device__ __forceinline__ void ctLoadX(const float* __restrict__ Xb, float& x0, float& x1, float& x2, float& x3) {
  constexpr unsigned nOut = SH::INC * SH::PIN;
  const unsigned t = threadIdx.x;
  x0 = Xb[t < nOut ? t : 0]; x1 = Xb[t + CT_NT < nOut ? t + CT_NT : 0];
  x2 = Xb[t + 2 * CT_NT < nOut ? t + 2 * CT_NT : 0]; x3 = Xb[t + 3 * CT_NT < nOut ? t + 3 * CT_NT : 0];
}
using CtA3 = CtShape<32, 64, 3, 3, 1, 3, 3>;      // 5 x 5 x 32 -> 64 k3 s1
using CtA2 = CtShape<16, 32, 4, 4, 1, 5, 5>;      // 8 x 8 x 16 -> 32 k4 s1
using CtA1 = CtShape<8, 16, 6, 6, 2, 8, 8>;       // 20 x 20 x 8 -> 16 k6 s2
constexpr int CT_ATARI_BUFD = 1536;                // floats: max(64 * 16, 32 * 48, 16 * 80)
constexpr int CT_ATARI_LDS = (2 * CT_ATARI_BUFD + 288 * 68) * 4;
static_assert(CtA3::KNC * CtA3::PD <= CT_ATARI_BUFD && CtA2::KNC * CtA2::PD <= CT_ATARI_BUFD && CtA1::KNC * CtA1::PD <= CT_ATARI_BUFD, "delta buffers");
static_assert(CtA1::NMT * 16 * CtA1::PT <= 288 * 68 && CtA2::NMT * 16 * CtA2::PT <= 288 * 68 && CtA3::NMT * 16 * CtA3::PT <= 288 * 68, "T buffer");
__global__ __launch_bounds__(CT_NT) void conv_back_atari_kernel(ConvArgs a) {      // a.nL == 4, layers 1 .. 3 of the shapes above
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  if (b >= a.B) return;
  CTSTAMP(0);
  float* sDa = reinterpret_cast<float*>(smem);      // deltas of layers 3 and 1
  float* sDb = sDa + CT_ATARI_BUFD;                  // deltas of layer 2
  float* sT = sDb + CT_ATARI_BUFD;
  const int tid = threadIdx.x;
  const float* W = a.W;
  float av[2][16];
  ctLoadAT<CtA3>(W + a.L[3].indW, av);
  {      // the deltas of the last layer (written by the dense layers' input-gradient launch): [c][9] -> [c][16]
    const float* src = a.L[3].D + (size_t)b * a.L[3].ldOut;
    for (unsigned e = tid; e < (unsigned)(CtA3::KNC * CtA3::P); e += CT_NT) { const unsigned c = e / (unsigned)CtA3::P, p = e - c * CtA3::P; sDa[c * CtA3::PD + p] = src[e]; }
  }
  float x0, x1, x2, x3;
  const float* X2 = a.L[2].X + (size_t)b * a.L[2].ldOut; float* D2 = a.L[2].D + (size_t)b * a.L[2].ldOut;
  const float* X1 = a.L[1].X + (size_t)b * a.L[1].ldOut; float* D1 = a.L[1].D + (size_t)b * a.L[1].ldOut;
  const float* X0 = a.L[0].X + (size_t)b * a.L[0].ldOut; float* D0 = a.L[0].D + (size_t)b * a.L[0].ldOut;
  ctLoadX<CtA3>(X2, x0, x1, x2, x3);
  __syncthreads();
  CTSTAMP(1);
  ctMfmaT<CtA3>(av, sDa, sT);
  ctLoadAT<CtA2>(W + a.L[2].indW, av);      // (the next layer's filter operand flies during this layer's col2im)
  CTSTAMP(2);
  __syncthreads();
  CTSTAMP(3);
  ctCol2imT<CtA3, true>(sT, sDb, D2, X2, x0, x1, x2, x3);
  ctLoadX<CtA2>(X1, x0, x1, x2, x3);
  CTSTAMP(4);
  __syncthreads();
  CTSTAMP(5);
  ctMfmaT<CtA2>(av, sDb, sT);
  ctLoadAT<CtA1>(W + a.L[1].indW, av);
  CTSTAMP(6);
  __syncthreads();
  CTSTAMP(7);
  ctCol2imT<CtA2, true>(sT, sDa, D1, X1, x0, x1, x2, x3);
  ctLoadX<CtA1>(X0, x0, x1, x2, x3);
  CTSTAMP(8);
  __syncthreads();
  CTSTAMP(9);
  ctMfmaT<CtA1>(av, sDa, sT);
  CTSTAMP(10);
  __syncthreads();
  CTSTAMP(11);
  ctCol2imT<CtA1, false>(sT, nullptr, D0, X0, x0, x1, x2, x3);
  CTSTAMP(12);
}
static bool ctIsShape(const ConvGeo& g, int inc, int knc, int ky, int kx, int s, int oy, int ox) {
  return g.InC == inc && g.KnC == knc && g.KnY == ky && g.KnX == kx && g.S == s && g.OpY == oy && g.OpX == ox;
}

// ---- conv_fwd_atari_kernel: the FORWARD pass of the same three layers as one launch (three launches of 5.7 us each until round 4; a
// first fused version in round 4 -- one wavefront per SIMD, run-time geometry, filters through LDS -- took the 20 us of the launches it
// replaced).  A workgroup of 16 wavefronts per row; per layer four (16 channels x 16 positions) tiles, each tile's reduction over the
// patch split over four wavefronts, joined in LDS in wave order (fixed summation order).  Layer_Conv2D.h:88-114:
//   X[c][p] = B[c][p] + sum_k K[c][k] in[patch_k(p)],  Y = SoftSign(X);      MFMA: M = channels, N = positions, reduction over the patch
// A operand = filter rows in the reference's layout, 16 bytes per lane and request (the reduction index is permuted inside groups of 16:
// lane group lc takes k = 16 S + 4 lc + j in sub-step j, for both operands); B operand = the row's input map in LDS through a table of
// patch offsets (one 16-byte LDS read per four steps).  The next layer's filter rows are requested in front of the join.
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <class SH> struct CtFwdGeo {
  static constexpr int NKS = SH::K / 16, MT = SH::KNC / 16, NT = SH::NNT, BASE = NKS / 4, REM = NKS % 4, MAXU = BASE + (REM ? 1 : 0);
  static_assert(SH::K % 16 == 0 && MT * NT == 4 && MAXU <= 5, "four tiles per layer, at most five groups of 16 patch elements per wavefront");
};
template <class SH>
__device__ __forceinline__ void ctFwdLoadA(const float* __restrict__ Wl, f32x4 (&av)[5]) {
  using G = CtFwdGeo<SH>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lc = lane >> 4;
  const int tile = wave >> 2, kq = wave & 3, mt = tile / G::NT;
  const int start = kq * G::BASE + (kq < G::REM ? kq : G::REM), count = G::BASE + (kq < G::REM ? 1 : 0);
  const f32x4* src = reinterpret_cast<const f32x4*>(Wl + (size_t)(mt * 16 + li) * SH::K + 16 * start + 4 * lc);
#pragma unroll
  for (int u = 0; u < G::MAXU; ++u) av[u] = src[u < count ? 4 * u : 0];
}
template <class SH, bool KEEP>
__device__ __forceinline__ void ctFwdLayerT(const f32x4 (&av)[5], const float* __restrict__ Bl, const float* __restrict__ sIn, const int* __restrict__ sK,
                                            float* __restrict__ sRed, float* __restrict__ sOut, float* __restrict__ Xg, float* __restrict__ Yg) {
  using G = CtFwdGeo<SH>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int tile = wave >> 2, kq = wave & 3, nt = tile % G::NT;
  const int start = kq * G::BASE + (kq < G::REM ? kq : G::REM), count = G::BASE + (kq < G::REM ? 1 : 0);
  // the join's element of this thread and its bias, requested now
  const int jt = tid >> 8, idx = tid & 255, jc = (jt / G::NT) * 16 + (idx >> 4), jp = (jt % G::NT) * 16 + (idx & 15);
  const bool jOk = jp < SH::P;
  const float bias = Bl[jOk ? jc * SH::P + jp : 0];
  // this lane's output position and the origin of its patch
  const int p = nt * 16 + li, pc = p < SH::P ? p : 0, oy = pc / SH::OPX, ox = pc - oy * SH::OPX;
  const float* inP = sIn + oy * SH::SS * SH::INX + ox * SH::SS;
  const i32x4* kt = reinterpret_cast<const i32x4*>(sK + 16 * start + 4 * lc);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < G::MAXU; ++u) {
    if (u < count) {
      const i32x4 ko = kt[4 * u];
      const float b0 = inP[ko[0]], b1 = inP[ko[1]], b2 = inP[ko[2]], b3 = inP[ko[3]];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], b1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], b2, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][3], b3, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sRed[wave * 256 + (4 * lc + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  const float* rp = sRed + jt * 1024 + idx;
  const float x = ((rp[0] + rp[256]) + (rp[512] + rp[768])) + bias;
  if (jOk) {
    const float y = x / (1 + fabsf(x));
    Xg[jc * SH::P + jp] = x; Yg[jc * SH::P + jp] = y;
    if (KEEP) sOut[jc * SH::P + jp] = y;
  }
}
template <class SH>
__device__ __forceinline__ void ctFwdTable(int* __restrict__ sK) {      // patch element k = (ic, fy, fx) -> offset in the input map [ic][iy][ix]
  for (int k = threadIdx.x; k < SH::K; k += CT_NT) { const int ic = k / SH::F, f = k - ic * SH::F, fy = f / SH::KNX, fx = f - fy * SH::KNX; sK[k] = ic * SH::PIN + fy * SH::INX + fx; }
}
constexpr int CT_FWD_LDS = (CtA1::INC * CtA1::PIN + CtA2::INC * CtA2::PIN + CtA3::INC * CtA3::PIN + 16 * 256 + CtA1::K + CtA2::K + CtA3::K) * 4;
__global__ __launch_bounds__(CT_NT) void conv_fwd_atari_kernel(ConvArgs a) {      // a.nL == 4: layers 1 .. 3 of the RACER_atari stack
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int nRows = a.sc->nRows[a.parity];
  const float* W = a.W;
  f32x4 av[5];
  ctFwdLoadA<CtA1>(W + a.L[1].indW, av);      // (before the test on the row count: itself a load)
  if (b >= nRows) return;
  CTSTAMP(24);
  float* sIn1 = reinterpret_cast<float*>(smem);                  // [8][20][20]   outputs of layer 0
  float* sIn2 = sIn1 + CtA1::INC * CtA1::PIN;                     // [16][8][8]
  float* sIn3 = sIn2 + CtA2::INC * CtA2::PIN;                     // [32][5][5]
  float* sRed = sIn3 + CtA3::INC * CtA3::PIN;                     // [16][256]
  int* sK1 = reinterpret_cast<int*>(sRed + 16 * 256); int* sK2 = sK1 + CtA1::K; int* sK3 = sK2 + CtA2::K;
  const int tid = threadIdx.x;
  {      // the row's first-layer outputs: flat 16-byte copy
    const f32x4* src = reinterpret_cast<const f32x4*>(a.L[0].Y + (size_t)b * a.L[0].ldOut);
    constexpr int n4 = CtA1::INC * CtA1::PIN / 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tid < n4) v = src[tid];
    ctFwdTable<CtA1>(sK1); ctFwdTable<CtA2>(sK2); ctFwdTable<CtA3>(sK3);
    if (tid < n4) reinterpret_cast<f32x4*>(sIn1)[tid] = v;
    static_assert(n4 <= CT_NT, "one piece per thread");
  }
  __syncthreads();
  CTSTAMP(25);
  ctFwdLayerT<CtA1, true>(av, W + a.L[1].indB, sIn1, sK1, sRed, sIn2, a.L[1].X + (size_t)b * a.L[1].ldOut, a.L[1].Y + (size_t)b * a.L[1].ldOut);
  ctFwdLoadA<CtA2>(W + a.L[2].indW, av);
  CTSTAMP(26);
  __syncthreads();
  CTSTAMP(27);
  ctFwdLayerT<CtA2, true>(av, W + a.L[2].indB, sIn2, sK2, sRed, sIn3, a.L[2].X + (size_t)b * a.L[2].ldOut, a.L[2].Y + (size_t)b * a.L[2].ldOut);
  ctFwdLoadA<CtA3>(W + a.L[3].indW, av);
  CTSTAMP(28);
  __syncthreads();
  CTSTAMP(29);
  ctFwdLayerT<CtA3, false>(av, W + a.L[3].indB, sIn3, sK3, sRed, nullptr, a.L[3].X + (size_t)b * a.L[3].ldOut, a.L[3].Y + (size_t)b * a.L[3].ldOut);
  CTSTAMP(30);
}
hipError_t launch_conv_fwd_tail(const ConvArgs& a, const ConvTailPlan& pl, int maxRows, hipStream_t s) {
  if (!pl.atari) return hipErrorInvalidValue;
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_fwd_atari_kernel), (size_t)CT_FWD_LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_fwd_atari_kernel, dim3(maxRows), dim3(CT_NT), (size_t)CT_FWD_LDS, s, a);
  return hipGetLastError();
}

// the plan: which layers the sample-resident kernels serve and how their LDS is cut (false: conv.hip's per-layer launches stay)
bool conv_tail_plan(const ConvGeo* L, int nL, ConvTailPlan* pl) {
  *pl = ConvTailPlan{};
  if (nL < 2) return false;
  const long long budget = 144 * 1024 / 4;      // floats of LDS per workgroup (of gfx950's 160 KB)
  long long bufD = 0;
  for (int l = 1; l < nL; ++l) {
    const ConvGeo& g = L[l];
    if (g.KnC != 16 && g.KnC != 32 && g.KnC != 64) return false;
    if (g.S != 1 && g.S != 2 && g.S != 4 && g.S != 8) return false;
    bufD = std::max<long long>(bufD, (long long)g.KnC * ctPitchD(g.P));
  }
  long long maxT = 0;
  for (int l = 1; l < nL; ++l) {
    const ConvGeo& g = L[l];
    const long long F = (long long)g.KnY * g.KnX, PT = ctPitchT(g.P), room = budget - 2 * bufD;
    long long icPer = room / (F * PT);
    // (rows of T are written in tiles of 16: the last tile of a chunk may overhang by up to 15 rows)
    while (icPer > 0 && (((icPer * F + 15) / 16 * 16) * PT > room || (icPer * F + 15) / 16 > 2 * CT_NW)) --icPer;      // (two row tiles per wavefront and pass)
    if (icPer < 1) return false;
    icPer = std::min<long long>(icPer, g.InC);
    pl->icPer[l] = (int)icPer;
    maxT = std::max(maxT, ((icPer * F + 15) / 16 * 16) * PT);
  }
  pl->bufD = (int)bufD;
  pl->ldsBack = (int)((2 * bufD + maxT) * 4);
  pl->on = 1;
  pl->atari = nL == 4 && ctIsShape(L[3], 32, 64, 3, 3, 1, 3, 3) && ctIsShape(L[2], 16, 32, 4, 4, 1, 5, 5) && ctIsShape(L[1], 8, 16, 6, 6, 2, 8, 8);
  return true;
}

hipError_t launch_conv_back(const ConvArgs& a, const ConvTailPlan& pl, hipStream_t s) {
  if (pl.atari) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_back_atari_kernel), (size_t)CT_ATARI_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_back_atari_kernel, dim3(a.B), dim3(CT_NT), (size_t)CT_ATARI_LDS, s, a);
    return hipGetLastError();
  }
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_back_kernel), (size_t)pl.ldsBack);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_back_kernel, dim3(a.B), dim3(CT_NT), (size_t)pl.ldsBack, s, a, pl);
  return hipGetLastError();
}

}  // namespace hl
