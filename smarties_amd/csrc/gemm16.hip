// smarties_amd/csrc/gemm16.hip -- the MLP contractions of the learner update on fp32 MFMA.
//
// gemm16_kernel: one 16x16 output tile per 256-thread workgroup, the reduction dimension split
// over the 4 wavefronts (one per SIMD), each issuing v_mfma_f32_16x16x4_f32 on two independent
// accumulators.  The minibatch problems are tiny (M = batch = 256, N,K <= 256), so the kernel
// is built for LATENCY, not for tile reuse: 256 tiles -> one workgroup per CU, every operand
// byte is fetched by ONE round of 16-byte global loads per thread (all issued before the first
// use), staged in LDS in layouts whose fragment reads (ds_read_b32) are bank-conflict free
// (ROWS tile: leading dimension 258 == 2 mod 32; COLS tile: leading dimension 16), and the four
// partial tiles are reduced through LDS before a fused epilogue:
//   EPI_FWD  bias + activation (+ parametric residual)      BaseLayer::forward (Layer_Base.h:64-95),
//                                                            ParametricResidualLayer::forward (Layers.h:347-361)
//   EPI_DX   residual back-prop + activation derivative      Layer::backward dX (Layers.h:133-147),
//                                                            ParametricResidualLayer::backward (:363-393)
//   EPI_DW   weight/bias gradient (+ fused Adam update)      Layer::backward dW (Layers.h:164-187),
//                                                            Adam::step (Optimizer.cpp:61-108)
// One launch can carry several problems (table in device memory); all dW / bias / residual
// parameter gradients of a step are ONE launch.
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
    P.C[j] = g;
    if (P.adam) { AdamCoef c; c.eta = sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac; adamApply(c, g, P.adW, P.adM1, P.adM2, j); }
  }
}

// one 16x16 output tile (or 16 columns of a column reduction) of problem P
// FL >= 0: the flavor (and with it the epilogue kind) is known at compile time and the development
// ablation switches are compiled out -- the operand loads then form one straight-line batch instead
// of a chain of branches with a wait at every join
template <int ROLE, int FL = -1, bool RAW = false>      // RAW: `tile` is tm * tilesN + tn as given (the caller placed its workgroups itself)
__device__ __forceinline__ void gemmTile(const GemmProblem& P, int tile, unsigned char* smem, const DevScalars* __restrict__ sc,
                                         const AdamHyper& hyp, int nRowsDyn) {
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
    const float* pw = isW ? P.adW : P.adbW; const float* p1 = isW ? P.adM1 : P.adbM1; const float* p2 = isW ? P.adM2 : P.adbM2;
    e0 = pw[iw]; e1 = p1[iw]; e2 = p2[iw];
  }
  const bool aRows = (flavor != GEMM_W);   // A tile is 16 rows x k  (else k x 16)
  const bool bRows = (flavor == GEMM_X);   // B tile is 16 rows x k  (else k x 16)

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 va[4], vb[4];
  // k handled by each wave: power of two in {8,16,32,64}; staged chunk kcp = 4*kw in {32..256}
  auto chunkGeo = [&](int kb, int& kc, int& kw, int& sh) { kc = min(KC, P.K - kb); kw = 8; sh = 3; while (4 * kw < kc) { kw <<= 1; ++sh; } };
  // every global load of one chunk (<= 8 x 16 B per thread), issued before any use.  With more than one chunk (weight gradients
  // over batch x BPTT rows) the loads of chunk i+1 are issued right after chunk i is staged, so they fly during its MFMA loop.
  auto loadChunk = [&](int kb) {
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
      if (bRows) {   // GEMM_X: weight rows n0.., reduction along the row
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
  const int kBeg = ks * KC, kEnd = (FL < 0 && P.nSplit > 1) ? min(P.K, kBeg + KC) : P.K;
  loadChunk(kBeg);
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
  const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  GSTAMP(27);
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
    if (m < P.M - 1) {
      const size_t i = (size_t)m * P.ldc + n;
      P.C[i] = v;
      if (P.adam) { adamStep(ac, v, e0, e1, e2); P.adW[i] = e0; P.adM1[i] = e1; P.adM2[i] = e2; }
    } else {
      P.biasOut[n] = v;
      if (P.adam) { adamStep(ac, v, e0, e1, e2); P.adbW[n] = e0; P.adbM1[n] = e1; P.adbM2[n] = e2; }
    }
  } else {
    P.C[(size_t)m * P.ldc + n] = v;
  }
  GSTAMP(28);
}

// ROLE only names the instantiation (fwd0 / fwd / dx / dw) so that a kernel trace separates the four
// launches of a step; the code is identical.
constexpr int GEMM_LDS = (2 * 16 * LDR + 4 * 256) * 4;
template <int ROLE>
__global__ __launch_bounds__(256) void gemm16_kernel(const GemmProblem* __restrict__ probs, int nProbs,
                                                     const DevScalars* __restrict__ sc, AdamHyper hyp, ExtraArgs extra, ExtraArgs extra2) {
  // one LDS block, used either by a GEMM tile (two operand tiles + the cross-wave reduction
  // buffer) or by the tail code of the extra workgroup
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES];
  // horizontal fusion: workgroup 0 of the grid (dispatched first) runs a piece of the step tail
  const int nRiders = (extra.role ? 1 : 0) + (extra2.role ? 1 : 0);
  if ((int)blockIdx.x < nRiders) { runExtra(blockIdx.x == 0 ? extra : extra2, smem); return; }
  const int bid = blockIdx.x - nRiders;
  const int nRowsDyn = sc->nRows[hyp.parity];   // issued together with the problem-table fetch
  int p = 0;
  for (int i = 1; i < nProbs; ++i) if (bid >= probs[i].tileStart) p = i;
  const GemmProblem P = probs[p];
  gemmTile<ROLE>(P, bid - P.tileStart, smem, sc, hyp, nRowsDyn);
}

// ---------------------------------------------------------------------------------------------------------------
// All dense forward layers of a network off the fused path in ONE launch (round 3; before: one launch of ~7 us per layer,
// of which the kernel boundary is 4-5).  Work is placed as in fused.hip: a 16-row panel of the minibatch belongs to a group
// of HT workgroups (HT = column tiles of the widest layer) on blockIdx = const (mod 8), i.e. on one XCD sharing one L2
// (checked by hl_create's probe; otherwise the per-layer launches stay).  Layer by layer every workgroup computes its
// column tile of the panel with the common tile code, then the group meets at a counter barrier (plain stores acknowledged
// by the L2 before the arrival, bounded spin) -- a layer needs all columns of the one before, of ITS panel only.
// Blocks 0..7: riders (so that the panels keep blockIdx % 8 = XCD).
// ---------------------------------------------------------------------------------------------------------------
struct ChainArgs { int n; int idx[HL_MAX_HIDDEN]; int HT; unsigned* panelCtr; };
__global__ __launch_bounds__(256) void fwd_chain_kernel(const GemmProblem* __restrict__ probs, ChainArgs ch, const DevScalars* __restrict__ sc, AdamHyper hyp,
                                                        ExtraArgs extra, ExtraArgs extra2) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES];
  if (blockIdx.x < 8) {
    if (blockIdx.x == 0 && extra.role) runExtra(extra, smem);
    else if (blockIdx.x == 1 && extra2.role) runExtra(extra2, smem);
    return;
  }
  const int bid = blockIdx.x - 8, xcd = bid & 7, gi = bid >> 3;
  const int HT = ch.HT, panel = (gi / HT) * 8 + xcd, n = gi % HT;
  const int nRowsDyn = sc->nRows[hyp.parity];
  if (panel * 16 >= nRowsDyn) return;                    // (the whole group of a panel leaves together)
  for (int l = 0; l < ch.n; ++l) {
    const GemmProblem P = probs[ch.idx[l]];
    if (n < P.tilesN) gemmTile<GEMM_ROLE_FWD, -1, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRowsDyn);
    if (l + 1 == ch.n) break;
    __builtin_amdgcn_s_waitcnt(0);                       // vmcnt(0): this tile's stores are acknowledged by the L2
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* ctr = ch.panelCtr + panel * 32;
      const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (old / (unsigned)HT + 1u) * (unsigned)HT;
      int spins = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { const_cast<DevScalars*>(sc)->errFlag = 80; break; }   // never hang the GPU on a lost workgroup
      }
    }
    __syncthreads();
  }
}
size_t fwd_chain_lds_bytes() { return GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES; }
int fwd_chain_blocks(int maxRows, int HT) { const int panels = (maxRows + 15) / 16, pg = (panels + 7) / 8; return 8 + 8 * HT * pg; }
hipError_t launch_fwd_chain(const GemmProblem* dProbs, const int* idx, int nLayers, int HT, int maxRows, unsigned* panelCtr, const DevScalars* sc,
                            const AdamHyper& hyp, const ExtraArgs* extra, const ExtraArgs* extra2, hipStream_t s) {
  ChainArgs ch{}; ch.n = nLayers; ch.HT = HT; ch.panelCtr = panelCtr;
  for (int l = 0; l < nLayers; ++l) ch.idx[l] = idx[l];
  ExtraArgs ex{}, ex2{}; if (extra) ex = *extra; if (extra2) ex2 = *extra2;
  hipLaunchKernelGGL(fwd_chain_kernel, dim3(fwd_chain_blocks(maxRows, HT)), dim3(256), 0, s, dProbs, ch, sc, hyp, ex, ex2);
  return hipGetLastError();
}

// the weight-gradient launch of the fused path: the problem table travels in the kernel arguments
// (scalar loads from the kernarg segment instead of two dependent global round trips)
// (the two riders' arguments travel unpacked: two whole ExtraArgs records would push the kernel-argument segment past 4 KB)
__global__ __launch_bounds__(256) void dw_table_kernel(DwTable tbl, const DevScalars* __restrict__ sc, AdamHyper hyp, int postOn, PostArgs post,
                                                       int sampPhases, int helpers, SampleArgs samp) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES];
#ifdef HL_TAIL_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 73) const_cast<DevScalars*>(sc)->dbgT[29] = wall_clock64();
#endif
  // riders: the bookkeeping of this step, then the index -> (episode, step) search of the NEXT minibatch (drawn and sorted by the
  // rider of the fused kernel) and the workgroups that gather it -- the sampler's dependency chain is split over both kernels
  const int r1 = postOn ? 1 : 0, r2 = sampPhases ? 1 + helpers : 0, nRiders = r1 + r2;
  if ((int)blockIdx.x < nRiders) {
    const int b = blockIdx.x;
    if (b < r1) postPhase(post, smem);
    else if (b == r1) samplePhases(samp, sampPhases, smem);
    else gatherHelper(samp, b - r1 - 1, helpers, smem);
    return;
  }
  const int bid = blockIdx.x - nRiders;
  int p = 0;
#pragma unroll
  for (int i = 1; i < DW_TABLE_MAX; ++i) if (i < tbl.n && bid >= tbl.p[i].tileStart) p = i;
  const GemmProblem P = tbl.p[p];      // by value: the whole record in one batch of scalar loads
  if (P.flavor == GEMM_W) gemmTile<GEMM_ROLE_DW, GEMM_W>(P, bid - P.tileStart, smem, sc, hyp, 0);
  else gemmTile<GEMM_ROLE_DW>(P, bid - P.tileStart, smem, sc, hyp, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward / dX tiles with a long reduction (256 < K <= 640: the dense layer behind a convolution stack, 576 -> 512, and
// wide first layers): the chunked tile above walks K in 256-column chunks, one global round trip each (11 us for
// 128 x 576 x 512).  Here EVERY operand load of the tile is in flight at once -- 2 x K / 64 16-byte loads per thread --,
// the whole reduction is staged in LDS once, the four wavefronts take a quarter of K each.  One problem per launch.
// ---------------------------------------------------------------------------------------------------------------
constexpr int OS_KMAX = 640, OS_Q = OS_KMAX / 64;       // float4 loads per thread and operand
__host__ __device__ inline size_t gemmOsLds(int K) { return ((size_t)2 * 16 * (K + 2) + 4 * 256) * 4; }
template <int ROLE>
__global__ __launch_bounds__(256) void gemm_os_kernel(const GemmProblem* __restrict__ probs, const DevScalars* __restrict__ sc, AdamHyper hyp, ExtraArgs extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char osmem[];
  if (extra.role && blockIdx.x == 0) { runExtra(extra, osmem); return; }
  const int bid = blockIdx.x - (extra.role ? 1 : 0);
  const int nRowsDyn = sc->nRows[hyp.parity];
  const GemmProblem P = probs[0];
  const bool isX = P.flavor == GEMM_X;
  int tile = bid;
  { const int nT = P.tilesM * P.tilesN; if ((nT & 7) == 0) tile = (tile & 7) * (nT >> 3) + (tile >> 3); }     // XCD-aware order (gemmTile)
  const int tm = tile / P.tilesN, tn = tile - tm * P.tilesN, m0 = tm * 16, n0 = tn * 16;
  const int Mvalid = P.dynRows ? nRowsDyn : P.M;
  if (m0 >= Mvalid) return;
  const int K = P.K, K4 = K >> 2, ld = K + 2;
  float* sA = reinterpret_cast<float*>(osmem);          // [16][K + 2]
  float* sB = sA + 16 * ld;                              // forward: [K][16]; dX: [16][K + 2]
  float* red = sB + 16 * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
  const bool outOk = m < Mvalid && n < P.N;
  // epilogue operands first (independent of everything)
  float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
  if (outOk) {
    if (!isX) { e0 = P.bias[n]; if (P.C3 && n < P.resN) { e1 = P.resIn[(size_t)m * P.ldRes + n]; e2 = P.resW[n]; e3 = P.resB[n]; } }
    else { if (n < P.resN) { e1 = P.resIn[(size_t)m * P.ldRes + n]; e2 = P.resW[n]; } e0 = P.actX[(size_t)m * P.ldAct + n]; e3 = P.actY[(size_t)m * P.ldAct + n]; }
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 va[OS_Q], vb[OS_Q];
  const int nA4 = 16 * K4;                                // float4 of a 16 x K rows tile == of a K x 16 columns tile
#pragma unroll
  for (int q = 0; q < OS_Q; ++q) {
    const int idx = tid + 256 * q; va[q] = z4; vb[q] = z4;
    if (idx < nA4) {
      const int r = idx / K4, c = (idx - r * K4) * 4;
      if (m0 + r < Mvalid) va[q] = *reinterpret_cast<const float4*>(P.A + (size_t)(m0 + r) * P.lda + c);
      if (isX) { if (n0 + r < P.N) vb[q] = *reinterpret_cast<const float4*>(P.B + (size_t)(n0 + r) * P.ldb + c); }
      else { const int k = idx >> 2, cc = n0 + (idx & 3) * 4; if (cc < P.ldb) vb[q] = *reinterpret_cast<const float4*>(P.B + (size_t)k * P.ldb + cc); }
    }
  }
#pragma unroll
  for (int q = 0; q < OS_Q; ++q) {
    const int idx = tid + 256 * q;
    if (idx < nA4) {
      const int r = idx / K4, c = (idx - r * K4) * 4;
      float2* d = reinterpret_cast<float2*>(sA + r * ld + c);
      d[0] = make_float2(va[q].x, va[q].y); d[1] = make_float2(va[q].z, va[q].w);
      if (isX) { float2* e = reinterpret_cast<float2*>(sB + r * ld + c); e[0] = make_float2(vb[q].x, vb[q].y); e[1] = make_float2(vb[q].z, vb[q].w); }
      else *reinterpret_cast<float4*>(sB + idx * 4) = vb[q];
    }
  }
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int kw = K >> 2, k0 = wave * kw;                  // K % 32 == 0 (checked by the launcher): kw is a multiple of 8
  for (int s = 0; s < kw; s += 8) {
    const int ka = k0 + s + lc, kb2 = ka + 4;
    const float a0 = sA[li * ld + ka], a1 = sA[li * ld + kb2];
    const float b0 = isX ? sB[li * ld + ka] : sB[ka * 16 + li], b1 = isX ? sB[li * ld + kb2] : sB[kb2 * 16 + li];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  if (!outOk) return;
  if (!isX) {
    const float x = v + e0;
    P.C[(size_t)m * P.ldc + n] = x;
    const float y = actEval(P.func, x);
    P.C2[(size_t)m * P.ldc + n] = y;
    if (P.C3) { float r = y; if (n < P.resN) r += e1 * e2 + e3; P.C3[(size_t)m * P.ldc + n] = r; }
  } else {
    float dres = v;
    if (n < P.resN) dres += e1 * e2;
    P.C[(size_t)m * P.ldc + n] = dres;
    P.C2[(size_t)m * P.ldc + n] = dres * actDiff(P.func, e0, e3);
  }
}
// usable for this problem?  (host side: K of the problem the launch carries)
bool gemm_oneshot_ok(int flavor, int K) { return (flavor == GEMM_F || flavor == GEMM_X) && K > 256 && K <= OS_KMAX && (K & 31) == 0; }
hipError_t launch_gemm_oneshot(int role, const GemmProblem* dProb, int K, int nBlocks, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s) {
  ExtraArgs ex{}; if (extra) ex = *extra;
  size_t lds = gemmOsLds(K); if (lds < TAIL_LDS_BYTES) lds = TAIL_LDS_BYTES;
  const dim3 grid(nBlocks + (ex.role ? 1 : 0)), block(256);
  const void* k = role == GEMM_ROLE_DX ? reinterpret_cast<const void*>(gemm_os_kernel<GEMM_ROLE_DX>) : (role == GEMM_ROLE_FWD0 ? reinterpret_cast<const void*>(gemm_os_kernel<GEMM_ROLE_FWD0>) : reinterpret_cast<const void*>(gemm_os_kernel<GEMM_ROLE_FWD>));
  { hipError_t e = ensureDynLds(k, lds); if (e != hipSuccess) return e; }
  if (role == GEMM_ROLE_DX) hipLaunchKernelGGL(gemm_os_kernel<GEMM_ROLE_DX>, grid, block, lds, s, dProb, sc, hyp, ex);
  else if (role == GEMM_ROLE_FWD0) hipLaunchKernelGGL(gemm_os_kernel<GEMM_ROLE_FWD0>, grid, block, lds, s, dProb, sc, hyp, ex);
  else hipLaunchKernelGGL(gemm_os_kernel<GEMM_ROLE_FWD>, grid, block, lds, s, dProb, sc, hyp, ex);
  return hipGetLastError();
}

// sum of the chunk partials of the split weight-gradient problems, in chunk order, + Adam
// (one more row of the grid, blockIdx.y == nProbs: its first workgroup is a rider -- the far-policy count + beta update a POST_DEFER
// bookkeeping pass of the dW launch left over)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmProblem* __restrict__ probs, const DevScalars* __restrict__ sc, AdamHyper hyp, int nProbs, PostArgs farBeta) {
  if ((int)blockIdx.y == nProbs) {
    __shared__ __attribute__((aligned(16))) unsigned char sFar[64];
    if (blockIdx.x == 0) farBetaPhase(farBeta, sFar);
    return;
  }
  const GemmProblem P = probs[blockIdx.y];
  if (P.nSplit <= 1) return;
  const bool col = P.flavor == RED_COL;         // column sums: one row of N values
  const int i = blockIdx.x * 256 + threadIdx.x, MN = col ? P.N : P.M * P.N;
  if (i >= MN) return;
  float s = 0.f;
  for (int c0 = 0; c0 < P.nSplit; c0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c0 + u < P.nSplit ? P.part[(size_t)(c0 + u) * MN + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  const int m = i / P.N, n = i - m * P.N;
  AdamCoef c; c.eta = sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac;
  if (col) {
    P.C[i] = s;
    if (P.adamRed) adamApply(c, s, P.adW, P.adM1, P.adM2, i);
  } else if (m < P.M - 1) {
    const size_t o = (size_t)m * P.ldc + n;
    P.C[o] = s;
    if (P.adamRed) adamApply(c, s, P.adW, P.adM1, P.adM2, o);
  } else {
    P.biasOut[n] = s;
    if (P.adamRed) adamApply(c, s, P.adbW, P.adbM1, P.adbM2, n);
  }
}
hipError_t launch_splitk_reduce(const GemmProblem* dProbs, int nProbs, int maxMN, const DevScalars* sc, const AdamHyper& hyp, hipStream_t s, const PostArgs* farBeta) {
  PostArgs fb{}; if (farBeta) fb = *farBeta;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((maxMN + 255) / 256, nProbs + (farBeta ? 1 : 0)), dim3(256), 0, s, dProbs, sc, hyp, nProbs, fb);
  return hipGetLastError();
}

hipError_t launch_dw_table(const DwTable& tbl, int nBlocks, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s,
                           const ExtraArgs* extra2) {
  PostArgs post{}; SampleArgs samp{}; int postOn = 0, phases = 0, helpers = 0;
  if (extra && extra->role == 2) { post = extra->post; postOn = 1; }
  if (extra2 && extra2->role == 1) { samp = extra2->samp; phases = extra2->phases; helpers = extra2->helpers; }
  hipLaunchKernelGGL(dw_table_kernel, dim3(nBlocks + postOn + (phases ? 1 + helpers : 0)), dim3(256), 0, s, tbl, sc, hyp, postOn, post, phases, helpers, samp);
  return hipGetLastError();
}

hipError_t launch_gemm(int role, const GemmProblem* dProbs, int nProbs, int nBlocks, const DevScalars* sc,
                       const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s, const ExtraArgs* extra2) {
  if (nBlocks <= 0) return hipSuccess;
  ExtraArgs ex{}, ex2{}; if (extra) ex = *extra; if (extra2) ex2 = *extra2;
  if (!ex.role && ex2.role) { ex = ex2; ex2 = ExtraArgs{}; }
  const dim3 grid(nBlocks + (ex.role ? 1 : 0) + (ex2.role ? 1 : 0)), block(256);
  switch (role) {
    case GEMM_ROLE_FWD0: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_FWD0>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    case GEMM_ROLE_FWD: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_FWD>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    case GEMM_ROLE_DX: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_DX>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    default: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_DW>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
  }
  return hipGetLastError();
}

}  // namespace hl
