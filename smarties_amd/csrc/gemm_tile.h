// smarties_amd/csrc/gemm_tile.h -- the 16x16 output tile of the MLP contractions (one workgroup of 256 threads, the reduction
// split over its four wavefronts on v_mfma_f32_16x16x4_f32) with its fused epilogues: shared by the per-layer launches, the
// forward chain (gemm16.hip).  See gemm16.hip for the layout rationale.
#pragma once
#include "tail_dev.h"

namespace hl {

#define KC 256
#define LDR 258

// development time stamps of one W1-gradient tile (-DHL_TAIL_STAMPS), DevScalars::dbgT[24..]
#ifdef HL_TAIL_STAMPS
#define GSTAMP(i) do { if (ROLE == GEMM_ROLE_DW && threadIdx.x == 0 && P.M > 200 && P.N > 200 && tile == 40) const_cast<DevScalars*>(sc)->dbgT[i] = wall_clock64(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void adamApply(const AdamCoef& c, float g, float* W, float* M1, float* M2, size_t i) {
  float w = W[i], m1 = M1[i], m2 = M2[i];
  adamStep(c, g, w, m1, m2);
  W[i] = w; M1[i] = m1; M2[i] = m2;
}

// Stores into a replica window (peer's or own) that leave this XCD's L2 behind for certain: system-scope write-through (sc0 sc1).  The
// windows are uncached memory and a plain store normally goes around the L2 as well -- but where the window's memory had an earlier life
// as cached memory (hipMalloc / hipFree of other learners in the process), lines of it may still sit in an L2, a plain store then HITS
// there, and a consumer inside the SAME launch (the folded exchange's chunk workgroups on another XCD) reads the old bytes from HBM:
// replicas a few ulps apart after some hundred steps, only behind other tests in one process (round 6).  A release fence per tile would
// do too -- and costs 15 us per step.
__device__ __forceinline__ void stWindow16(unsigned char* p, const f32x4& x) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void stWindow4(unsigned char* p, float x) {
  __hip_atomic_store(reinterpret_cast<float*>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// this replica's slot in a peer's window for the collective the gradient belongs to (byte offset inside the window)
__device__ __forceinline__ size_t pushSlot(const PushArgs& pu) {
  const unsigned long long seq = __hip_atomic_load(&pu.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (size_t)pu.slotsOffset + ((size_t)(seq & 1) * (size_t)pu.nRanks + (size_t)pu.rank) * (size_t)pu.slotBytes;
}

__device__ __forceinline__ void redcol_tile(const GemmProblem& P, int tile, float* red, const DevScalars* sc,
                                            const AdamHyper& hyp, int ks) {
  // out[j] = sum_m A[m][j] * (B ? B[m][j] : 1): 16 columns per workgroup, 16 row-partitions,
  // 8 independent loads in flight per thread
  const int tid = threadIdx.x, jj = tid & 15, part = tid >> 4;
  const int j = tile * 16 + jj;
  float acc = 0.f;
  // split problems (many rows: batch x BPTT steps): this workgroup sums the 256 rows of chunk ks only
  const int mBeg = P.nSplit > 1 ? ks * KC : 0, mEnd = P.nSplit > 1 ? min(P.K, mBeg + KC) : P.K;
  if (j < P.N) {
    for (int m0 = mBeg + part; m0 < mEnd; m0 += 128) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + 16 * u;
        av[u] = m < mEnd ? P.A[(size_t)m * P.lda + j] : 0.f;
        bv[u] = (P.B && m < mEnd) ? P.B[(size_t)m * P.ldb + j] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += av[u] * bv[u];
    }
  }
  red[part * 16 + jj] = acc;
  __syncthreads();
  if (part == 0 && j < P.N) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) g += red[q * 16 + jj];
    if (P.nSplit > 1) { P.part[(size_t)ks * P.N + j] = g; return; }
    if (!(hyp.push.on && hyp.push.self)) P.C[j] = g;      // (folded replica launch: windows only, see gemmTile)
    if (hyp.push.on) {      // replicas: into the peers' windows too (a handful of columns: element stores)
      const size_t so = pushSlot(hyp.push) + (size_t)((P.C - hyp.push.gBase) + j) * 4;
      for (int p = 0; p < hyp.push.nRanks; ++p) if (p != hyp.push.rank || hyp.push.self) stWindow4(hyp.push.peers[p] + so, g);
    }
    if (P.adam) { AdamCoef c; c.eta = sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac; adamApply(c, g, P.adW, P.adM1, P.adM2, j); }
  }
}

// one 16x16 output tile (or 16 columns of a column reduction) of problem P
// FL >= 0: the flavor (and with it the epilogue kind) is known at compile time and the development
// ablation switches are compiled out -- the operand loads then form one straight-line batch instead
// of a chain of branches with a wait at every join
// the B operand (weights) of a tile's first chunk, requested ahead of the tile: chained stages (mlp_panel.hip) issue these loads in
// front of the group barrier the stage waits at -- the weights depend on nothing the stage before computes
struct TileB { float4 v[4]; };
__device__ __forceinline__ void gemmLoadB(const GemmProblem& P, int tileRaw, TileB& out) {
  const int flavor = P.flavor, tid = threadIdx.x;
  const int kRem = (flavor == GEMM_F || flavor == GEMM_X) && P.K > KC && (P.K & (KC - 1)) != 0 && (P.K & (KC - 1)) <= 4 ? (P.K & (KC - 1)) : 0;
  const int Kmain = P.K - kRem, kc = min(KC, Kmain);
  int kw = 8, sh = 3; while (4 * kw < kc) { kw <<= 1; ++sh; }
  const int nf4 = kw, n0 = (tileRaw % P.tilesN) * 16;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    out.v[q] = z4;
    if (flavor == GEMM_X) {
      const int r = idx >> sh, c = (idx & (nf4 - 1)) * 4;
      if (idx < 16 * nf4 && n0 + r < P.N && c < kc && c < P.ldb) out.v[q] = *reinterpret_cast<const float4*>(P.B + (size_t)(n0 + r) * P.ldb + c);
    } else {
      const int k = idx >> 2, c = n0 + (idx & 3) * 4;
      if (idx < 16 * nf4 && k < kc && c < P.ldb && c < ((P.N + 3) & ~3)) out.v[q] = *reinterpret_cast<const float4*>(P.B + (size_t)k * P.ldb + c);
    }
  }
}

template <int ROLE, int FL = -1, bool RAW = false>      // RAW: `tile` is tm * tilesN + tn as given (the caller placed its workgroups itself)
__device__ __forceinline__ void gemmTile(const GemmProblem& P, int tile, unsigned char* smem, const DevScalars* __restrict__ sc,
                                         const AdamHyper& hyp, int nRowsDyn, const TileB* preB = nullptr) {
  const int flavor = FL >= 0 ? FL : P.flavor;
#ifdef HL_DEV
  const int variant = FL >= 0 ? 0 : hyp.variant;      // development ablation switches (HL_EXTRA_FLAGS=-DHL_DEV)
#else
  constexpr int variant = 0;
#endif
  const int epi = FL == GEMM_W ? EPI_DW : P.epi;
  float* sA = reinterpret_cast<float*>(smem);
  float* sB = sA + 16 * LDR;
  float* red = sB + 16 * LDR;
  int ks = 0;                                  // chunk of the reduction (split problems only)
  if (FL < 0 && P.nSplit > 1) { const int nT0 = P.tilesM * P.tilesN; ks = tile / nT0; tile -= ks * nT0; }
  if (flavor == RED_COL) { redcol_tile(P, tile, red, sc, hyp, ks); return; }
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.
  // Give XCD x the contiguous (row-major) tile range [x*nT/8, (x+1)*nT/8): the tiles of one XCD
  // then share their A row-panels, and each L2 fetches 1/8 of A instead of all of it.
  {
    const int nT = P.tilesM * P.tilesN;
    if (!RAW && (nT & 7) == 0 && !(variant & 16) && !((variant >> 5) & (1 << ROLE))) tile = (tile & 7) * (nT >> 3) + (tile >> 3);
  }

  const int tm = tile / P.tilesN, tn = tile - tm * P.tilesN;
  const int m0 = tm * 16, n0 = tn * 16;
  const int Mvalid = P.dynRows ? nRowsDyn : P.M;
  if (m0 >= Mvalid) return;
  if (variant & 8) return;              // ablation: launch + problem-table fetch only

  GSTAMP(24);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lc = lane >> 4;
  const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
  const bool outOk = m < Mvalid && n < P.N;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
  AdamCoef ac{};
  if (outOk && !(variant & 4)) {        // ablation: no epilogue prefetch
    if (epi == EPI_FWD) {
      e0 = P.bias[n];
      if (P.C3 && n < P.resN) { e1 = P.resIn[(size_t)m * P.ldRes + n]; e2 = P.resW[n]; e3 = P.resB[n]; }
    } else if (epi == EPI_DX) {
      if (n < P.resN) { e1 = P.resIn[(size_t)m * P.ldRes + n]; e2 = P.resW[n]; }
      e0 = P.actX[(size_t)m * P.ldAct + n]; e3 = P.actY[(size_t)m * P.ldAct + n];
    } else if (epi == EPI_DW && P.adam && FL < 0) {
      ac.eta = sc->etaEff[hyp.parity]; ac.lambda = hyp.lambda; ac.fac = hyp.fac;
      if (m < P.M - 1) { const size_t i = (size_t)m * P.ldc + n; e0 = P.adW[i]; e1 = P.adM1[i]; e2 = P.adM2[i]; }
      else { e0 = P.adbW[n]; e1 = P.adbM1[n]; e2 = P.adbM2[n]; }
    }
  }
  if (FL == GEMM_W && P.adam) {   // branch-free variant: a divergent if/else here ends in a wait for its loads
    ac.eta = sc->etaEff[hyp.parity]; ac.lambda = hyp.lambda; ac.fac = hyp.fac;
    const bool isW = m < P.M - 1;
    const size_t iw = outOk ? (isW ? (size_t)m * P.ldc + n : (size_t)n) : 0;
    const float* pw = pickPtr(isW, P.adW, P.adbW); const float* p1 = pickPtr(isW, P.adM1, P.adbM1); const float* p2 = pickPtr(isW, P.adM2, P.adbM2);
    e0 = pw[iw]; e1 = p1[iw]; e2 = p2[iw];
  }
  // a reduction one to four elements longer than a whole number of 256-element chunks (257 observed states: the Humanoid wrapper)
  // would cost a chunk round of its own -- a dependent global round trip, two barriers -- for those few columns: they are taken by
  // the element threads instead (operands requested here, with the epilogue's), the chunk loop stops in front of them
  const int kRem = (flavor == GEMM_F || flavor == GEMM_X) && P.K > KC && (P.K & (KC - 1)) != 0 && (P.K & (KC - 1)) <= 4 ? (P.K & (KC - 1)) : 0;
  const int Kmain = P.K - kRem;
  float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
  if (kRem && outOk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < kRem) {
      const int k = Kmain + q;
      ra[q] = P.A[(size_t)m * P.lda + k];
      rb[q] = flavor == GEMM_X ? P.B[(size_t)n * P.ldb + k] : P.B[(size_t)k * P.ldb + n];
    }
  }
  const bool aRows = (flavor != GEMM_W);   // A tile is 16 rows x k  (else k x 16)
  const bool bRows = (flavor == GEMM_X);   // B tile is 16 rows x k  (else k x 16)

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 va[4], vb[4];
  // k handled by each wave: power of two in {8,16,32,64}; staged chunk kcp = 4*kw in {32..256}
  auto chunkGeo = [&](int kb, int& kc, int& kw, int& sh) { kc = min(KC, Kmain - kb); kw = 8; sh = 3; while (4 * kw < kc) { kw <<= 1; ++sh; } };
  // every global load of one chunk (<= 8 x 16 B per thread), issued before any use.  With more than one chunk (weight gradients
  // over batch x BPTT rows) the loads of chunk i+1 are issued right after chunk i is staged, so they fly during its MFMA loop.
  auto loadChunk = [&](int kb, bool havB = false) {
    int kc, kw, sh; chunkGeo(kb, kc, kw, sh);
    const int nf4 = kw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      va[q] = z4; vb[q] = z4;
      if (variant & 1) continue;        // ablation: no operand loads
      if (aRows) {
        const int r = idx >> sh, c = (idx & (nf4 - 1)) * 4;
        if (idx < 16 * nf4 && m0 + r < Mvalid && c < kc && kb + c < P.lda)
          va[q] = *reinterpret_cast<const float4*>(P.A + (size_t)(m0 + r) * P.lda + kb + c);
      } else {   // GEMM_W: rows = reduction (batch), columns m0.. = input features, + the ones column
        const int k = idx >> 2, c = m0 + (idx & 3) * 4;
        // (the ones column is patched in at staging time: touching the value here would make the
        // compiler wait for every load before issuing the next one)
        if (idx < 16 * nf4 && k < kc && c < P.lda) va[q] = *reinterpret_cast<const float4*>(P.A + (size_t)(kb + k) * P.lda + c);
      }
      if (havB) vb[q] = preB->v[q];      // (requested by the caller ahead of this call: gemmLoadB)
      else if (bRows) {   // GEMM_X: weight rows n0.., reduction along the row
        const int r = idx >> sh, c = (idx & (nf4 - 1)) * 4;
        if (idx < 16 * nf4 && n0 + r < P.N && c < kc && kb + c < P.ldb)
          vb[q] = *reinterpret_cast<const float4*>(P.B + (size_t)(n0 + r) * P.ldb + kb + c);
      } else {
        const int k = idx >> 2, c = n0 + (idx & 3) * 4;
        if (idx < 16 * nf4 && k < kc && c < P.ldb && c < ((P.N + 3) & ~3))
          vb[q] = *reinterpret_cast<const float4*>(P.B + (size_t)(kb + k) * P.ldb + c);
      }
    }
  };
  const int kBeg = ks * KC, kEnd = (FL < 0 && P.nSplit > 1) ? min(P.K, kBeg + KC) : Kmain;
  loadChunk(kBeg, preB != nullptr);
  for (int kb = kBeg; kb < kEnd; kb += KC) {
    int kc, kw, sh; chunkGeo(kb, kc, kw, sh);
    const int nf4 = kw;                       // float4 per 16-row-tile row (= kcp/4)
    GSTAMP(30);
    // ---- stage into LDS ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      if (idx < 16 * nf4) {
        if (aRows) {
          const int r = idx >> sh, c = (idx & (nf4 - 1)) * 4;
          float2* d = reinterpret_cast<float2*>(sA + r * LDR + c);
          d[0] = make_float2(va[q].x, va[q].y); d[1] = make_float2(va[q].z, va[q].w);
        } else {
          float4 v = va[q];
          const int k = idx >> 2, c = m0 + (idx & 3) * 4;
          const int one = P.M - 1 - c;       // position of the ones column inside this float4
          if (k < kc) { if (one == 0) v.x = 1.f; else if (one == 1) v.y = 1.f; else if (one == 2) v.z = 1.f; else if (one == 3) v.w = 1.f; }
          if (one < 0) v = z4;
          else { if (one < 1) v.y = 0.f; if (one < 2) v.z = 0.f; if (one < 3) v.w = 0.f; }
          *reinterpret_cast<float4*>(sA + idx * 4) = v;           // [k][16]
        }
        if (bRows) {
          const int r = idx >> sh, c = (idx & (nf4 - 1)) * 4;
          float2* d = reinterpret_cast<float2*>(sB + r * LDR + c);
          d[0] = make_float2(vb[q].x, vb[q].y); d[1] = make_float2(vb[q].z, vb[q].w);
        } else {
          *reinterpret_cast<float4*>(sB + idx * 4) = vb[q];
        }
      }
    }
    __syncthreads();
    GSTAMP(25);
    if (kb + KC < kEnd) loadChunk(kb + KC);
    const int k0 = wave * kw;
    if (!(variant & 2))                 // ablation: no MFMA loop
    for (int s = 0; s < kw; s += 8) {
      const int ka = k0 + s + lc, kb2 = ka + 4;
      const float a0 = aRows ? sA[li * LDR + ka] : sA[ka * 16 + li];
      const float b0 = bRows ? sB[li * LDR + ka] : sB[ka * 16 + li];
      const float a1 = aRows ? sA[li * LDR + kb2] : sA[kb2 * 16 + li];
      const float b1 = bRows ? sB[li * LDR + kb2] : sB[kb2 * 16 + li];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }
  GSTAMP(26);
  // ---- cross-wave reduction of the 4 partial tiles ----
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
#pragma unroll
  for (int q = 0; q < 4; ++q) if (q < kRem) v = fmaf(ra[q], rb[q], v);
  GSTAMP(27);
  if (epi == EPI_DW && hyp.push.on && !(FL < 0 && P.nSplit > 1)) {
    // replicas connected through peer windows: the tile goes into every peer's window as 16-byte stores (regrouped through LDS:
    // a lane holds one element, a 16-byte store wants four of a row); columns beyond N carry zeros, like the padding of G
    __syncthreads();                         // every thread has read the four partial tiles
    red[tid] = outOk ? v : 0.f;
    __syncthreads();
    if (tid < 64) {
      const int r = tid >> 2, c = (tid & 3) * 4, mm = m0 + r, nn = n0 + c;
      const bool isW = mm < P.M - 1, isB = mm == P.M - 1;
      const int lim = isW ? P.ldc : ((P.N + 7) & ~7);
      if ((isW || isB) && nn < lim) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(red + r * 16 + c);
        const size_t off = isW ? (size_t)(P.C - hyp.push.gBase) + (size_t)mm * P.ldc + nn : (size_t)(P.biasOut - hyp.push.gBase) + nn;
        const size_t so = pushSlot(hyp.push) + off * 4;
        for (int p = 0; p < hyp.push.nRanks; ++p) if (p != hyp.push.rank || hyp.push.self) stWindow16(hyp.push.peers[p] + so, x);
        __builtin_amdgcn_s_waitcnt(0);          // acknowledged before this wavefront ends
      }
    }
  }
  if (!outOk) return;

  if (epi == EPI_FWD) {
    const float x = v + e0;
    P.C[(size_t)m * P.ldc + n] = x;
    const float y = actEval(P.func, x);
    P.C2[(size_t)m * P.ldc + n] = y;
    if (P.C3) {
      float r = y;
      if (n < P.resN) r += e1 * e2 + e3;
      P.C3[(size_t)m * P.ldc + n] = r;
    }
  } else if (epi == EPI_DX) {
    float dres = v;
    if (n < P.resN) dres += e1 * e2;
    P.C[(size_t)m * P.ldc + n] = dres;
    P.C2[(size_t)m * P.ldc + n] = dres * actDiff(P.func, e0, e3);
  } else if (epi == EPI_DW && FL < 0 && P.nSplit > 1) {
    P.part[((size_t)ks * P.M + m) * P.N + n] = v;
  } else if (epi == EPI_DW) {
    {      // weight rows and the bias row (the ones row of the product) through ONE store sequence on selected pointer values: as two
           // branches the compiler merged their stores and indexed the record's pointers on the stack (scratch)
      const bool isW = m < P.M - 1;
      const size_t i = isW ? (size_t)m * P.ldc + n : (size_t)n;
      // (folded replica launch: the gradient lives in the windows only -- the chunk workgroups write the SUM to this array from other
      //  XCDs, and a local tile left dirty in this XCD's L2 could be written back on top of it)
      if (!(hyp.push.on && hyp.push.self)) pickPtrW(isW, P.C, P.biasOut)[i] = v;
      if (P.adam) { adamStep(ac, v, e0, e1, e2); pickPtrW(isW, P.adW, P.adbW)[i] = e0; pickPtrW(isW, P.adM1, P.adbM1)[i] = e1; pickPtrW(isW, P.adM2, P.adbM2)[i] = e2; }
    }
  } else {
    P.C[(size_t)m * P.ldc + n] = v;
  }
  GSTAMP(28);
}

// ROLE only names the instantiation (fwd0 / fwd / dx / dw) so that a kernel trace separates the four
// launches of a step; the code is identical.
constexpr int GEMM_LDS = (2 * 16 * LDR + 4 * 256) * 4;

}  // namespace hl
