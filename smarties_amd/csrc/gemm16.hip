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
#include "dw_wide_dev.h"
#include "head_body.h"
#include "xchg_dev.h"

namespace hl {

template <int ROLE>
__global__ __launch_bounds__(256) void gemm16_kernel(const GemmProblem* __restrict__ probs, int nProbs,
                                                     const DevScalars* __restrict__ sc, AdamHyper hyp, ExtraArgs extra, ExtraArgs extra2) {
  // one LDS block, used either by a GEMM tile (two operand tiles + the cross-wave reduction
  // buffer) or by the tail code of the extra workgroup
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES];
  // horizontal fusion: workgroup 0 of the grid (dispatched first) runs a piece of the step tail
  // ... and, behind the riders, the workgroups that gather the minibatch a sampler rider found (PH_PUBLISH: extra.helpers of them)
  const int nRid = (extra.role ? 1 : 0) + (extra2.role ? 1 : 0), nHelp = extra.role == 1 ? extra.helpers : 0, nRiders = nRid + nHelp;
  if ((int)blockIdx.x < nRid) { runExtra(blockIdx.x == 0 ? extra : extra2, smem); return; }
  if ((int)blockIdx.x < nRiders) { gatherHelper(extra.samp, blockIdx.x - nRid, nHelp, smem); return; }
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

// ---------------------------------------------------------------------------------------------------------------
// The whole step in front of the weight gradients in ONE launch, for dense networks off the fused kernels (three or more hidden layers,
// unequal widths: settings/RACER_glider.json is 3 x 128 under the Gaussian advantage): the forward chain above, then the head of the
// panel's 16 samples (head_body.h: four per workgroup, or one per workgroup above 128 units), then the input-gradient products from the
// last hidden layer down -- every stage a column tile per workgroup, the panel's group meeting at the counter barrier in between.  What a
// stage reads was written by workgroups of its own group (one XCD, one L2): plain stores, acknowledged before the arrival.
// Five launches (forward chain, head, a dX launch per layer, dW) become two.  Blocks 0..7: riders as in fused.hip (block 0: draws and
// sort of the next minibatch); its index search, the gather and this step's bookkeeping ride the dW launch (launchWeightGrad).
// ---------------------------------------------------------------------------------------------------------------
struct StepChainArgs { int nF, nX; int fIdx[HL_MAX_HIDDEN], xIdx[HL_MAX_HIDDEN]; int HT; unsigned* panelCtr; };
__device__ __forceinline__ void chainBarrier(unsigned* ctr, int HT, const DevScalars* sc) {
  __builtin_amdgcn_s_waitcnt(0);                       // vmcnt(0): this workgroup's stores are acknowledged by the L2
  __syncthreads();
  if (threadIdx.x == 0) {
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
template <int HQ, int SPLIT>
__global__ __launch_bounds__(256) void step_chain_kernel(const GemmProblem* __restrict__ probs, StepChainArgs ch, HeadArgs ha, const DevScalars* __restrict__ sc,
                                                         AdamHyper hyp, ExtraArgs extra) {
  constexpr int LDS0 = GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS0 > HEAD_LDS ? LDS0 : HEAD_LDS];
  if (blockIdx.x < 8) {
    if (blockIdx.x == 0 && extra.role) runExtra(extra, smem);
    else if (blockIdx.x == 1 && ha.deferBeta) farBetaPhase(extra.post, smem);      // (the count and beta the bookkeeping of the step before left over)
    return;
  }
  const int bid = blockIdx.x - 8, xcd = bid & 7, gi = bid >> 3;
  const int HT = ch.HT, panel = (gi / HT) * 8 + xcd, n = gi % HT;
  const int nRowsDyn = sc->nRows[hyp.parity];
  if (panel * 16 >= nRowsDyn) return;                    // (the whole group of a panel leaves together)
  unsigned* ctr = ch.panelCtr + panel * 32;
  // development time stamps of workgroup (panel 0, tile 0), 100 MHz clock: -DHL_CHAIN_STAMPS (tools/chain_stamps.py)
#ifdef HL_CHAIN_STAMPS
#define CSTMP(i) do { if (threadIdx.x == 0 && panel == 0 && n == 0) const_cast<DevScalars*>(sc)->dbgT[i] = wall_clock64(); } while (0)
#else
#define CSTMP(i) do { } while (0)
#endif
  CSTMP(0);
  for (int l = 0; l < ch.nF; ++l) {
    const GemmProblem P = probs[ch.fIdx[l]];
    if (n < P.tilesN) gemmTile<GEMM_ROLE_FWD, -1, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRowsDyn);
    CSTMP(1 + 2 * l);
    chainBarrier(ctr, HT, sc);
    CSTMP(2 + 2 * l);
  }
  // the panel's samples: rows 4 g + wave of workgroup g (SPLIT = 1) / row g (SPLIT = 4), further ones by the same workgroup where the
  // group is smaller than that
  if constexpr (SPLIT == 1) {
    for (int g = n; g < 4; g += HT) headBody<HQ, 1>(ha, panel * 16 + 4 * g + (int)(threadIdx.x >> 6), smem);
  } else {
    for (int g = n; g < 16; g += HT) { headBody<HQ, 4>(ha, panel * 16 + g, smem); __syncthreads(); }
  }
  CSTMP(16);
  for (int l = 0; l < ch.nX; ++l) {
    chainBarrier(ctr, HT, sc);
    CSTMP(17 + 2 * l);
    const GemmProblem P = probs[ch.xIdx[l]];
    if (n < P.tilesN) gemmTile<GEMM_ROLE_DX, -1, true>(P, panel * P.tilesN + n, smem, sc, hyp, nRowsDyn);
    CSTMP(18 + 2 * l);
  }
}
hipError_t launch_step_chain(const GemmProblem* dProbs, const int* fIdx, int nF, const int* xIdx, int nX, int HT, int maxRows, unsigned* panelCtr,
                             const HeadArgs& ha, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s) {
  StepChainArgs ch{}; ch.nF = nF; ch.nX = nX; ch.HT = HT; ch.panelCtr = panelCtr;
  for (int l = 0; l < nF; ++l) ch.fIdx[l] = fIdx[l];
  for (int l = 0; l < nX; ++l) ch.xIdx[l] = xIdx[l];
  ExtraArgs ex{}; if (extra) ex = *extra;
  const dim3 grid(fwd_chain_blocks(maxRows, HT)), block(256);
  if (ha.H > 128) {
    const int HQ = (ha.H + 255) / 256;
    if (HQ <= 1) hipLaunchKernelGGL((step_chain_kernel<1, 4>), grid, block, 0, s, dProbs, ch, ha, sc, hyp, ex);
    else if (HQ <= 2) hipLaunchKernelGGL((step_chain_kernel<2, 4>), grid, block, 0, s, dProbs, ch, ha, sc, hyp, ex);
    else return hipErrorInvalidValue;
  } else if (ha.H > 64) hipLaunchKernelGGL((step_chain_kernel<2, 1>), grid, block, 0, s, dProbs, ch, ha, sc, hyp, ex);
  else hipLaunchKernelGGL((step_chain_kernel<1, 1>), grid, block, 0, s, dProbs, ch, ha, sc, hyp, ex);
  return hipGetLastError();
}

// One 16 x 16 weight-gradient tile of problem P (flavor GEMM_W, rows = the minibatch) with its whole reduction in this workgroup of four
// wavefronts, operands straight from memory into the MFMA: no LDS staging and no barrier in front of the cross-wave reduction
// (the staged form: gemm_tile.h).  Epilogue as gemmTile's EPI_DW without the peer-window push.
__device__ __forceinline__ void dwDirectTile(const GemmProblem& P, int tile, unsigned char* smem, const DevScalars* sc, const AdamHyper& hyp) {
  float* red = reinterpret_cast<float*>(smem);
  {   // XCD-aware tile order, as gemmTile: the tiles of one XCD share their A row-panels
    const int nT = P.tilesM * P.tilesN;
    if ((nT & 7) == 0) tile = (tile & 7) * (nT >> 3) + (tile >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = tile / P.tilesN, tn = tile - tm * P.tilesN;
  const int m0 = tm * 16, n0 = tn * 16;
  const int li = lane & 15, lc = lane >> 4;
  const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
  const bool outOk = m < P.M && n < P.N;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
  AdamCoef ac{};
  if (P.adam) {
    ac.eta = sc->etaEff[hyp.parity]; ac.lambda = hyp.lambda; ac.fac = hyp.fac;
    const bool isW = m < P.M - 1;
    const size_t iw = outOk ? (isW ? (size_t)m * P.ldc + n : (size_t)n) : 0;
    const float* pw = pickPtr(isW, P.adW, P.adbW); const float* p1 = pickPtr(isW, P.adM1, P.adbM1); const float* p2 = pickPtr(isW, P.adM2, P.adbM2);
    e0 = pw[iw]; e1 = p1[iw]; e2 = p2[iw];
  }
  const int RW = (((P.K + 3) / 4) + 3) & ~3;      // rows per wavefront: a multiple of four
  const int r0 = wave * RW, rEnd = min(P.K, r0 + RW);
  const int ma = m0 + li, nb = n0 + li;
  const bool aOne = ma == P.M - 1, aOk = ma < P.M - 1 && ma < P.lda, bOk = nb < P.N;
  const float* pA = P.A + (aOk ? ma : 0);
  const float* pB = P.B + (bOk ? nb : 0);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (RW <= 32) dwwRows<8>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);
  else if (RW <= 64) dwwRows<16>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);
  else dwwRows<34>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  if (!outOk) return;
  {      // (one store sequence on selected pointer values: gemm_tile.h)
    const bool isW = m < P.M - 1;
    const size_t i = isW ? (size_t)m * P.ldc + n : (size_t)n;
    pickPtrW(isW, P.C, P.biasOut)[i] = v;
    if (P.adam) { adamStep(ac, v, e0, e1, e2); pickPtrW(isW, P.adW, P.adbW)[i] = e0; pickPtrW(isW, P.adM1, P.adbM1)[i] = e1; pickPtrW(isW, P.adM2, P.adbM2)[i] = e2; }
  }
}

// the weight-gradient launch of the fused path: the problem table travels in the kernel arguments
// (scalar loads from the kernarg segment instead of two dependent global round trips)
// (the two riders' arguments travel unpacked: two whole ExtraArgs records would push the kernel-argument segment past 4 KB)
// The launch's body.  FOLD (replicas over peer windows, SMARTIES_HIP_FOLD=1): the exchange of the gradient this launch produces is part of
// the launch -- the tiles store into every window (the own one too) and count themselves, the bookkeeping rider pushes the counters
// message, and fold.nCh chunk workgroups at the END of the grid do what the exchange launch does (xchg_dev.h).  Two kernels, so that the
// step of ONE learner keeps the kernel it had (with the fold's arguments and branches in it, dw_table_kernel took 7.0 instead of 6.6 us).
template <bool FOLD>
__device__ __forceinline__ void dwTableBody(const DwTable& tbl, const DevScalars* __restrict__ sc, const AdamHyper& hyp, int postOn, const PostArgs& post,
                                            int sampPhases, int helpers, const SampleArgs& samp, const FoldArgs& fold) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > TAIL_LDS_BYTES ? GEMM_LDS : TAIL_LDS_BYTES];
#ifdef HL_TAIL_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 73) const_cast<DevScalars*>(sc)->dbgT[29] = wall_clock64();
#endif
#ifdef HL_STEP_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 73) const_cast<DevScalars*>(sc)->dbgStep[64 + (sc->sampleSeq & 63)] = wall_clock64();      // (sampleSeq: stable inside this launch)
#endif
  // riders: the bookkeeping of this step, then the index -> (episode, step) search of the NEXT minibatch (drawn and sorted by the
  // rider of the fused kernel) and the workgroups that gather it -- the sampler's dependency chain is split over both kernels
  const int r1 = postOn ? 1 : 0, r2 = sampPhases ? 1 + helpers : 0, nRiders = r1 + r2;
  if ((int)blockIdx.x < nRiders) {
    const int b = blockIdx.x;
    if (b < r1) {
      if ((FOLD && fold.on)) FOSTAMP(sc, 0);
      postPhase(post, smem);
      if ((FOLD && fold.on)) FOSTAMP(sc, 1);
      if ((FOLD && fold.on)) {      // the counters message (sixteen floats thread 0 just wrote behind the gradient) into every window, then this producer's arrival
        __syncthreads();
        if (threadIdx.x < 16) {
          const float x = __hip_atomic_load(post.cntMsg + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const size_t so = pushSlot(hyp.push) + (size_t)((post.cntMsg - hyp.push.gBase) + threadIdx.x) * 4;
          for (int p = 0; p < hyp.push.nRanks; ++p) stWindow4(hyp.push.peers[p] + so, x);
        }
        // Everything this pass wrote goes to memory BEFORE it counts itself (an agent-scope release: this XCD's L2 is written back).  Not for
        // the readers' sake alone: the chunk workgroup that closes the step -- on whichever XCD -- writes some of the same words (the
        // summed counters over the local ones, DevScalars::cnt), and two L2s holding the same words dirty with different values leave
        // it to the order of their write-backs which one survives (seen: replicas drifting apart by one ulp of beta after 25 steps)
        __threadfence();
        foldArrive(fold.ctl, (unsigned)fold.nTiles + 1u);
        FOSTAMP(sc, 2);
      }
    }
    else if (b == r1) samplePhases(samp, sampPhases, smem);
    else gatherHelper(samp, b - r1 - 1, helpers, smem);
    return;
  }
  const int bid = blockIdx.x - nRiders;
  if ((FOLD && fold.on) && bid >= fold.nTiles) {      // chunk workgroups of the folded exchange
    XchgCore c; c.msg = fold.msg; c.n = fold.n; c.nRanks = hyp.push.nRanks; c.rank = hyp.push.rank; c.peers = hyp.push.peers;
    c.slotsOffset = (size_t)hyp.push.slotsOffset; c.slotBytes = (size_t)hyp.push.slotBytes; c.ctl = fold.ctl; c.sc = const_cast<DevScalars*>(sc);
    c.timeoutTicks = fold.timeoutTicks; c.pushed = fold.n; c.localTarget = (unsigned)fold.nTiles + 1u;
    XchgAdam ad; ad.W = fold.W; ad.M1 = fold.M1; ad.M2 = fold.M2; ad.n = fold.nAdam; ad.lambda = hyp.lambda; ad.fac = hyp.fac; ad.parity = hyp.parity;
    xchgChunk<float, true, true>(c, ad, post, POST_BETA, bid - fold.nTiles, fold.nCh, reinterpret_cast<XchgLds*>(smem));
    return;
  }
  if ((FOLD && fold.on) && bid == 40) FOSTAMP(sc, 3);
  int p = 0;
#pragma unroll
  for (int i = 1; i < DW_TABLE_MAX; ++i) if (i < tbl.n && bid >= tbl.p[i].tileStart) p = i;
  const GemmProblem P = tbl.p[p];      // by value: the whole record in one batch of scalar loads
  if (P.flavor == GEMM_W) {
#ifndef HL_DW_TABLE_STAGED
    if (!hyp.push.on && P.K <= 1024) { dwDirectTile(P, bid - P.tileStart, smem, sc, hyp); return; }      // (peer-window push: the staged form's epilogue)
#endif
    gemmTile<GEMM_ROLE_DW, GEMM_W>(P, bid - P.tileStart, smem, sc, hyp, 0);
  }
  else gemmTile<GEMM_ROLE_DW>(P, bid - P.tileStart, smem, sc, hyp, 0);
  if ((FOLD && fold.on) && bid == 40) FOSTAMP(sc, 4);
  if ((FOLD && fold.on)) foldArrive(fold.ctl, (unsigned)fold.nTiles + 1u);
  if ((FOLD && fold.on) && bid == 40) FOSTAMP(sc, 5);
  if ((FOLD && fold.on) && bid == fold.nTiles - 1) FOSTAMP(sc, 14);
}
__global__ __launch_bounds__(256) void dw_table_kernel(DwTable tbl, const DevScalars* __restrict__ sc, AdamHyper hyp, int postOn, PostArgs post,
                                                       int sampPhases, int helpers, SampleArgs samp) {
  dwTableBody<false>(tbl, sc, hyp, postOn, post, sampPhases, helpers, samp, FoldArgs{});
}
__global__ __launch_bounds__(256) void dw_table_fold_kernel(DwTable tbl, const DevScalars* __restrict__ sc, AdamHyper hyp, int postOn, PostArgs post,
                                                            int sampPhases, int helpers, SampleArgs samp, FoldArgs fold) {
  dwTableBody<true>(tbl, sc, hyp, postOn, post, sampPhases, helpers, samp, fold);
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
                           const ExtraArgs* extra2, const FoldArgs* fold) {
  PostArgs post{}; SampleArgs samp{}; int postOn = 0, phases = 0, helpers = 0;
  if (extra && extra->role == 2) { post = extra->post; postOn = 1; }
  if (extra2 && extra2->role == 1) { samp = extra2->samp; phases = extra2->phases; helpers = extra2->helpers; }
  FoldArgs fo{}; if (fold) fo = *fold;
  // (a folded launch needs its bookkeeping rider -- it produces the counters message and is one of the counted producers -- and windows
  //  that take the own values too)
  if (fo.on && (!postOn || !hyp.push.on || !hyp.push.self || !post.cntMsg || fo.nTiles != nBlocks || fo.nCh < 1 || fo.nCh > XCHG_CHUNKS)) return hipErrorInvalidValue;
  if (fo.on) hipLaunchKernelGGL(dw_table_fold_kernel, dim3(nBlocks + postOn + (phases ? 1 + helpers : 0) + fo.nCh), dim3(256), 0, s, tbl, sc, hyp, postOn, post, phases, helpers, samp, fo);
  else hipLaunchKernelGGL(dw_table_kernel, dim3(nBlocks + postOn + (phases ? 1 + helpers : 0)), dim3(256), 0, s, tbl, sc, hyp, postOn, post, phases, helpers, samp);
  return hipGetLastError();
}

hipError_t launch_gemm(int role, const GemmProblem* dProbs, int nProbs, int nBlocks, const DevScalars* sc,
                       const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s, const ExtraArgs* extra2) {
  if (nBlocks <= 0 && !(extra && extra->role) && !(extra2 && extra2->role)) return hipSuccess;      // (riders keep the launch alive)
  ExtraArgs ex{}, ex2{}; if (extra) ex = *extra; if (extra2) ex2 = *extra2;
  if (!ex.role && ex2.role) { ex = ex2; ex2 = ExtraArgs{}; }
  if (ex.role != 1 && ex2.role == 1) { const ExtraArgs t = ex; ex = ex2; ex2 = t; }      // (a sampler rider with helpers goes first)
  if (ex2.role == 1) ex2.helpers = 0;
  if (ex.role != 1) ex.helpers = 0;
  const dim3 grid(nBlocks + (ex.role ? 1 : 0) + (ex2.role ? 1 : 0) + ex.helpers), block(256);
  switch (role) {
    case GEMM_ROLE_FWD0: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_FWD0>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    case GEMM_ROLE_FWD: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_FWD>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    case GEMM_ROLE_DX: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_DX>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
    default: hipLaunchKernelGGL(gemm16_kernel<GEMM_ROLE_DW>, grid, block, 0, s, dProbs, nProbs, sc, hyp, ex, ex2); break;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients over MANY rows (recurrent nets: batch x BPTT steps, 2176 at the shipped shape) in ONE launch, without the
// (tile, 256-row chunk) workgroups + splitk_reduce_kernel of the common launch: a 16 x 16 tile belongs to DW_WIDE_Q = 4 workgroups of
// four wavefronts, one per quarter of the rows; a wavefront takes a fourth of its quarter with its operands straight from memory
// into the MFMA (rows are the reduction index: lane (li, lc) of step s reads A[r + 4 s + lc][m0 + li] and D[r + 4 s + lc][n0 + li]
// -- 64-byte row pieces, no LDS staging), every load in flight at once; the four partial tiles meet in LDS in wave order, the
// quarter's tile goes to memory, and the LAST of the four workgroups to arrive (a counter per tile, never reset: arrival 4 n + 3)
// sums the four in quarter order and runs gemmTile's epilogue (gradient, fused Adam).  The hand-off goes through the coherence
// point (agent-scope stores and loads, as the fused kernel's safe mode): it holds wherever the four workgroups run.
// RED_COL problems (column sums over the same rows): the same four quarters, 16 row partitions each.  Riders as gemm16_kernel (256 threads).
// ---------------------------------------------------------------------------------------------------------------
// nq: workgroups per tile (1: a tile's whole reduction in one workgroup -- minibatch rows only, no hand-off; DW_WIDE_Q: row quarters)
// (body: dw_wide_dev.h)
__global__ __launch_bounds__(DWW_NT) void dw_wide_kernel(const GemmProblem* __restrict__ probs, int nProbs, int nTiles, int nq, float* __restrict__ part,
                                                         unsigned* __restrict__ ctr, const DevScalars* __restrict__ sc, AdamHyper hyp,
                                                         ExtraArgs extra, ExtraArgs extra2) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DWW_LDS > TAIL_LDS_BYTES ? DWW_LDS : TAIL_LDS_BYTES];
  const int nRid = (extra.role ? 1 : 0) + (extra2.role ? 1 : 0);
  if ((int)blockIdx.x < nRid) { runExtra(blockIdx.x == 0 ? extra : extra2, smem); return; }
  dwWideBody(probs, nProbs, nTiles, nq, part, ctr, sc, hyp, (int)blockIdx.x - nRid, smem);
}
// nq = 1: part / ctr are not used (may be null)
hipError_t launch_dw_wide(const GemmProblem* dProbs, int nProbs, int nTiles, int nq, float* part, unsigned* ctr, const DevScalars* sc, const AdamHyper& hyp,
                          const ExtraArgs* extra, const ExtraArgs* extra2, hipStream_t s) {
  if (nTiles <= 0) return hipSuccess;
  if (hyp.push.on || (nq != 1 && nq != DW_WIDE_Q) || (nq > 1 && (!part || !ctr))) return hipErrorInvalidValue;      // (replicas pushing tiles into peer windows keep the common launch)
  ExtraArgs ex{}, ex2{}; if (extra) ex = *extra; if (extra2) ex2 = *extra2;
  if (!ex.role && ex2.role) { ex = ex2; ex2 = ExtraArgs{}; }
  ex.helpers = 0; ex2.helpers = 0;
  const int nBlocks = ((nTiles + 7) / 8) * 8 * nq;
  hipLaunchKernelGGL(dw_wide_kernel, dim3(nBlocks + (ex.role ? 1 : 0) + (ex2.role ? 1 : 0)), dim3(DWW_NT), 0, s, dProbs, nProbs, nTiles, nq, part, ctr, sc, hyp, ex, ex2);
  return hipGetLastError();
}

}  // namespace hl
