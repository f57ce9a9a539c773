// smarties_amd/csrc/bigmm.hip -- dense layers at large local batches (1024 < B <= 16384), where the step is no latency chain any
// more but 2 B x 0.42 MFLOP of fp32 matrix products (Network/Layers/Layer_Base.h:64-113 forward / backward, as gemm_tile.h).
//
//   big_panel_kernel   forward  C = f(A W + b) [+ parametric residual]   and   dX   Dres = D W^T [+ residual], D_below = Dres f'.
//                      WEIGHT-STATIONARY: a workgroup stages a 64-column tile of W (all K <= 256 rows of it: 68 KB) in LDS once and
//                      walks over the 64-row panels of its share of the minibatch; each of its four wavefronts owns 16 rows of a
//                      panel -- their K values straight from HBM / L2 into registers as 16-byte loads, the next panel's in flight
//                      while this one multiplies -- and reads its B operands for FOUR column tiles with one ds_read_b128
//                      (W is stored [k][16 columns x 4 tiles]).  No cross-wave reduction, no barrier inside the loop.
//                      Per 16 x 64 x K block: K/4 x 4 MFMAs (16x16x4 fp32) against K/4 LDS reads and K/16 global loads per lane.
//   big_dw_kernel      weight gradients  G[m][n] = sum_rows A[row][m] D[row][n]  (+ the bias row): 64 x 64 output tiles, the rows cut
//                      into chunks (about 640 workgroups per problem, one per (tile, chunk)); 32-row slices of A and D go through LDS (k-major,
//                      pitch 80: the four row groups of an MFMA operand fall into different banks), each wavefront owns a 32 x 32
//                      quadrant.  Partial tiles land where splitk_reduce_kernel (gemm16.hip) expects them: summed in chunk order
//                      there, with the Adam update.
// MFMA operand convention as in gemm_tile.h: lane (li = lane & 15, lc = lane >> 4) feeds A[row li][k lc] and B[k lc][col li] and
// receives C[row 4 lc + i][col li], i = 0..3.  The k index of a step may be ANY assignment as long as A and B agree: step (j, c)
// of lane group lc is k = 16 j + 4 lc + c, which is what a lane's j-th 16-byte load of its row holds.
#include "dev_common.h"

namespace hl {

constexpr int BP_PITCH = 64;      // floats per k-row of the staged weight tile (a quarter wavefront reads 256 contiguous bytes: no padding needed)
constexpr int BP_OUT = 16 * 64;    // floats of a wavefront's output tile in LDS (the epilogue's transposition)

template <int KP, bool TRANSW>
__global__ __launch_bounds__(256, 2) void big_panel_kernel(GemmProblem P, const DevScalars* __restrict__ sc, int parity, int nGroups) {
  extern __shared__ __attribute__((aligned(16))) float sW[];      // [16 KP][BP_PITCH] + [4 wavefronts][BP_OUT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  // workgroup -> (row group, column tile): the column tiles of one row group read the same rows of A, so they sit on ONE XCD
  // (workgroup b runs on XCD b mod 8) and share its L2 -- otherwise every XCD fetches all of A
  const int colTiles = (P.N + 63) / 64, slot = (int)blockIdx.x >> 3;
  const int rowGroup = ((int)blockIdx.x & 7) + 8 * (slot / colTiles);
  if (rowGroup >= nGroups) return;
  const int n0 = (slot % colTiles) * 64;
  const int nRows = P.dynRows ? sc->nRows[parity] : P.M;
  constexpr int KPAD = 16 * KP;
  // ---- the weight tile, once: sW[k][4 (n & 15) + (n >> 4)] = Wop[k][n0 + n], zeros outside K x N ----
  // (every load of the tile is requested before the first LDS write: a rolled load -> store loop pays one memory round trip per
  //  iteration, 16 of them at K = 256 -- 23 us per workgroup, measured)
  if constexpr (!TRANSW) {        // forward: W is [K][ldb], a row of it = the outputs of input k
    f32x4 w[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int idx = tid + 256 * q, k = idx >> 4, c4 = (idx & 15) * 4;
      w[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (k < P.K) {
        if (n0 + c4 + 3 < P.ldb) w[q] = *reinterpret_cast<const f32x4*>(P.B + (size_t)k * P.ldb + n0 + c4);
        else for (int e = 0; e < 4; ++e) if (n0 + c4 + e < P.ldb) w[q][e] = P.B[(size_t)k * P.ldb + n0 + c4 + e];
      }
    }
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int idx = tid + 256 * q, k = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int n = c4 + e; sW[k * BP_PITCH + 4 * (n & 15) + (n >> 4)] = n0 + n < P.N ? w[q][e] : 0.f; }
    }
  } else {                        // dX: W is [N][ldb] (N = inputs of the layer), the reduction runs along its rows
    f32x4 w[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int idx = tid + 256 * q, n = idx / (KPAD / 4), k4 = (idx - n * (KPAD / 4)) * 4;
      w[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n0 + n < P.N) {
        if (k4 + 3 < P.ldb) w[q] = *reinterpret_cast<const f32x4*>(P.B + (size_t)(n0 + n) * P.ldb + k4);
        else for (int e = 0; e < 4; ++e) if (k4 + e < P.ldb) w[q][e] = P.B[(size_t)(n0 + n) * P.ldb + k4 + e];
      }
    }
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int idx = tid + 256 * q, n = idx / (KPAD / 4), k4 = (idx - n * (KPAD / 4)) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) sW[(k4 + e) * BP_PITCH + 4 * (n & 15) + (n >> 4)] = k4 + e < P.K ? w[q][e] : 0.f;
    }
  }
  // Epilogue through LDS: a lane holds 4 rows x 1 column of each of its four 16 x 16 tiles, stores of that shape are 64-byte
  // pieces (measured: the launch was bound by them).  The wavefront's 16 x 64 tile goes through its own 4 KB of LDS (columns
  // rotated by 16 per group of four rows: conflict-free writes) and leaves as 16-byte pieces of whole rows: lane l owns the four
  // columns 4 (l & 15) .. + 3 of the rows (l >> 4) + 4 q.  Their per-column operands:
  float* sOut = sW + KPAD * BP_PITCH + wave * BP_OUT;
  const int ec = 4 * (lane & 15);
  // what later launches read of a hidden layer: f'(x) takes the pre-activation OR the output (actDiff), the layers above and the weight
  // gradients the block output (C3 behind a parametric residual, else the output)
  const bool keepX = !actDiffFromOutput(P.func), keepY = !keepX || P.C3 == nullptr;
  const float* actSrc = keepX ? P.actX : P.actY;
  float eb[4], ew[4], er[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = n0 + ec + e;
    eb[e] = (!TRANSW && n < P.N) ? P.bias[n] : 0.f;
    ew[e] = (n < P.resN && P.resW) ? P.resW[n] : 0.f;
    er[e] = (!TRANSW && n < P.resN && P.resB) ? P.resB[n] : 0.f;
  }
  __syncthreads();
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto loadA = [&](f32x4 (&a)[KP], int pb) {
    const int row = (pb * 4 + wave) * 16 + li;
#pragma unroll
    for (int j = 0; j < KP; ++j) a[j] = row < nRows ? *reinterpret_cast<const f32x4*>(P.A + (size_t)row * P.lda + 16 * j + 4 * lc) : z4;
  };
  f32x4 aCur[KP], aNxt[KP];
  int pb = rowGroup;
  if (pb * 64 < nRows) loadA(aCur, pb);
  for (; pb * 64 < nRows; pb += nGroups) {
    const bool more = (pb + nGroups) * 64 < nRows;
    if (more) loadA(aNxt, pb + nGroups);
    f32x4 acc[4] = {z4, z4, z4, z4};
    // the LDS reads of step group j + 1 are issued in front of the 16 MFMAs of group j (the scheduling barrier keeps the compiler
    // from hoisting ALL reads of the panel in front of the first MFMA: 128 more registers, spills)
    f32x4 bCur[4], bNxt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bCur[c] = *reinterpret_cast<const f32x4*>(sW + (4 * lc + c) * BP_PITCH + 4 * li);
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (j + 1 < KP) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bNxt[c] = *reinterpret_cast<const f32x4*>(sW + (16 * (j + 1) + 4 * lc + c) * BP_PITCH + 4 * li);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float av = aCur[j][c];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bCur[c][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bCur[c][1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bCur[c][2], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bCur[c][3], acc[3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < KP) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bCur[c] = bNxt[c];
      }
    }
    // ---- epilogue ----
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) sOut[(4 * lc + i) * 64 + ((16 * t + li + 16 * lc) & 63)] = acc[t][i];
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the wavefront's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int rBase = (pb * 4 + wave) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = (lane >> 4) + 4 * q, m = rBase + rr, n = n0 + ec;
      const f32x4 v = *reinterpret_cast<const f32x4*>(sOut + rr * 64 + ((ec + 16 * (rr >> 2)) & 63));
      if (m >= nRows || n >= P.N) continue;
      const size_t o = (size_t)m * P.ldc + n;
      const bool whole = n + 3 < P.N;
      if constexpr (!TRANSW) {
        f32x4 rin = z4;
        if (P.C3 && n < P.resN) { if (n + 3 < P.ldRes) rin = *reinterpret_cast<const f32x4*>(P.resIn + (size_t)m * P.ldRes + n); }
        f32x4 x, y, r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = v[e] + eb[e]; y[e] = actEval(P.func, x[e]); r[e] = n + e < P.resN ? y[e] + (rin[e] * ew[e] + er[e]) : y[e]; }
        if (whole) {
          if (keepX) *reinterpret_cast<f32x4*>(P.C + o) = x;
          if (keepY) *reinterpret_cast<f32x4*>(P.C2 + o) = y;
          if (P.C3) *reinterpret_cast<f32x4*>(P.C3 + o) = r;
        } else for (int e = 0; e < 4; ++e) if (n + e < P.N) { if (keepX) P.C[o + e] = x[e]; if (keepY) P.C2[o + e] = y[e]; if (P.C3) P.C3[o + e] = r[e]; }
      } else {
        f32x4 rin = z4, ax = z4;
        if (n < P.resN && n + 3 < P.ldRes) rin = *reinterpret_cast<const f32x4*>(P.resIn + (size_t)m * P.ldRes + n);
        if (n + 3 < P.ldAct) ax = *reinterpret_cast<const f32x4*>(actSrc + (size_t)m * P.ldAct + n);
        f32x4 dres, d;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dres[e] = n + e < P.resN ? v[e] + rin[e] * ew[e] : v[e]; d[e] = dres[e] * actDiff(P.func, ax[e], ax[e]); }
        if (whole) { *reinterpret_cast<f32x4*>(P.C + o) = dres; *reinterpret_cast<f32x4*>(P.C2 + o) = d; }
        else for (int e = 0; e < 4; ++e) if (n + e < P.N) { P.C[o + e] = dres[e]; P.C2[o + e] = d[e]; }
      }
    }
    __builtin_amdgcn_wave_barrier();               // (the next panel's tile overwrites sOut)
    if (more) {
#pragma unroll
      for (int j = 0; j < KP; ++j) aCur[j] = aNxt[j];
    }
  }
}

template <int KP, bool TRANSW> static hipError_t bigPanelLaunch(const GemmProblem& P, const DevScalars* sc, int parity, hipStream_t s) {
  const size_t lds = ((size_t)16 * KP * BP_PITCH + 4 * BP_OUT) * sizeof(float);
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(big_panel_kernel<KP, TRANSW>), lds); if (e != hipSuccess) return e;
  const int colTiles = (P.N + 63) / 64;
  int groups = std::max(1, 512 / colTiles);                   // two workgroups per CU
  groups = std::min(groups, (P.M + 63) / 64);
  hipLaunchKernelGGL((big_panel_kernel<KP, TRANSW>), dim3(8 * ((groups + 7) / 8) * colTiles), dim3(256), lds, s, P, sc, parity, groups);
  return hipGetLastError();
}
bool big_panel_ok(const GemmProblem& P) {
  const int K = P.K, kp = (K + 15) / 16, kpad = 16 * (kp <= 2 ? 2 : (kp <= 4 ? 4 : (kp <= 8 ? 8 : 16)));      // (the instantiation's K)
  // (16-byte row loads up to the padded K: the rows of A are at least that long -- zeros or finite values behind K -- and 16-byte aligned)
  // (16-byte epilogue accesses: every row pitch a multiple of 4, the residual input and the activations as long as the output rows)
  const bool al = (P.lda & 3) == 0 && (P.ldb & 3) == 0 && (P.ldc & 3) == 0 && (!P.resN || ((P.ldRes & 3) == 0 && P.ldRes >= ((P.resN + 3) & ~3))) &&
                  (P.flavor != GEMM_X || ((P.ldAct & 3) == 0 && P.ldAct >= ((P.N + 3) & ~3)));
  return (P.flavor == GEMM_F || P.flavor == GEMM_X) && K <= 256 && kpad <= P.lda && al;
}
hipError_t launch_big_panel(const GemmProblem& P, const DevScalars* sc, int parity, hipStream_t s) {
  const int kp = (P.K + 15) / 16;
  const bool tr = P.flavor == GEMM_X;
#define BP_CASE(KPV) (tr ? bigPanelLaunch<KPV, true>(P, sc, parity, s) : bigPanelLaunch<KPV, false>(P, sc, parity, s))
  if (kp <= 2) return BP_CASE(2);
  if (kp <= 4) return BP_CASE(4);
  if (kp <= 8) return BP_CASE(8);
  return BP_CASE(16);
#undef BP_CASE
}

// ---------------------------------------------------------------------------------------------------------------------
// big_mm_kernel: the same two products as 64 x 64 output tiles with BOTH operands in 32-deep k-slices through LDS (double-buffered:
// the next slice's 16-byte loads are in flight while this one multiplies), each wavefront a 32 x 32 quadrant -- the weight-gradient
// kernel's structure.  39 KB of LDS: four workgroups per CU, whose load / MFMA / epilogue phases overlap (the panel kernel above
// has two with two panels each: its phases add up).  A slice [64 rows][32 k] at pitch 36 (rows 16 apart in a tile fall into
// different banks with the four k groups); W slices [32 k][64 n] at pitch 80 (forward) or [64 n][32 k] at pitch 36 (dX: W rows are
// the outputs there).  The epilogue goes through LDS as in the panel kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MM_PB = 80;
// development time stamps of one tile's workgroup of the K > 64 forward product (-DHL_BIGMM_STAMPS; tools/bigmm_stamps.py)
#ifdef HL_BIGMM_STAMPS
#define MMSTAMP(i) do { if (!TRANSW && P.K > 64 && threadIdx.x == 0 && blockIdx.x == 40) const_cast<DevScalars*>(sc)->dbgT[i] = wall_clock64(); } while (0)
#define MMEND() do { if (!TRANSW && P.K > 64) { __syncthreads(); if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&const_cast<DevScalars*>(sc)->dbgT[10]), (unsigned long long)wall_clock64()); } } while (0)
#else
#define MMSTAMP(i) do { } while (0)
#define MMEND() do { } while (0)
#endif
// MM_KS: depth of a k-slice.  32: 39 KB of LDS, four workgroups per CU (grids of many tiles: their phases overlap); 64: half as many
// load -> barrier round trips per tile, 76 KB, two per CU -- for grids of at most two tiles per CU, where a tile's own chain is the launch
template <bool TRANSW, int MM_KS>
__global__ __launch_bounds__(256, MM_KS == 32 ? 4 : 2) void big_mm_kernel(GemmProblem P, const DevScalars* __restrict__ sc, int parity, int nRowTiles) {
  constexpr int MM_PA = MM_KS + 4, PPR = MM_KS / 4, RPP = 256 / PPR, Q = PPR / 4;      // row pitch of a slice; 16-byte pieces per row, rows per pass, passes
  __shared__ __attribute__((aligned(16))) float sA[2][64 * MM_PA];
  __shared__ __attribute__((aligned(16))) float sB[2][TRANSW ? 64 * MM_PA : MM_KS * MM_PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int colTiles = (P.N + 63) / 64, slot = (int)blockIdx.x >> 3;
  const int rowTile = ((int)blockIdx.x & 7) + 8 * (slot / colTiles);      // (the column tiles of a row tile on one XCD: they read the same rows of A)
  if (rowTile >= nRowTiles) return;
  MMSTAMP(0);
  const int m0 = rowTile * 64, n0 = (slot % colTiles) * 64;
  const int nRows = P.dynRows ? sc->nRows[parity] : P.M;
  if (m0 >= nRows) return;
  MMSTAMP(1);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int K = P.K;
  f32x4 va[Q], vb[Q];
  // a slice's loads touch nothing but their destination registers -- addresses clamped into the arrays, what lies outside the operands
  // zeroed when the slice is stored: with guards or masks at the load the compiler waited for the data in front of the slice's
  // products (s_waitcnt vmcnt(0) right behind the requests: one exposed memory round trip per slice, 1.15 us of it stamped)
  auto loadSlice = [&](int k0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      { const int r = min(m0 + tid / PPR + RPP * q, nRows - 1), k = min(k0 + (tid % PPR) * 4, P.lda - 4);
        va[q] = *reinterpret_cast<const f32x4*>(P.A + (size_t)r * P.lda + k); }
      if constexpr (TRANSW) {      // W rows n0 .. n0 + 63, columns k
        const int n = min(n0 + tid / PPR + RPP * q, P.N - 1), k = min(k0 + (tid % PPR) * 4, P.ldb - 4);
        vb[q] = *reinterpret_cast<const f32x4*>(P.B + (size_t)n * P.ldb + k);
      } else {                     // W rows k, columns n0 .. n0 + 63
        const int k = min(k0 + (tid >> 4) + 16 * q, K - 1), c = min(n0 + (tid & 15) * 4, P.ldb - 4);
        vb[q] = *reinterpret_cast<const f32x4*>(P.B + (size_t)k * P.ldb + c);
      }
    }
  };
  auto storeSlice = [&](int buf, int k0) {      // (k0: the slice the registers hold)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      { const int r = m0 + tid / PPR + RPP * q, k = k0 + (tid % PPR) * 4;
        f32x4 v = va[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (r >= nRows || k + e >= K) v[e] = 0.f;
        *reinterpret_cast<f32x4*>(&sA[buf][(tid / PPR + RPP * q) * MM_PA + (tid % PPR) * 4]) = v; }
      f32x4 w = vb[q];
      if constexpr (TRANSW) {
        const int n = n0 + tid / PPR + RPP * q, k = k0 + (tid % PPR) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n >= P.N || k + e >= K) w[e] = 0.f;
        *reinterpret_cast<f32x4*>(&sB[buf][(tid / PPR + RPP * q) * MM_PA + (tid % PPR) * 4]) = w;
      } else {
        const int k = k0 + (tid >> 4) + 16 * q, c = n0 + (tid & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k >= K || c + e >= P.N) w[e] = 0.f;
        *reinterpret_cast<f32x4*>(&sB[buf][((tid >> 4) + 16 * q) * MM_PB + (tid & 15) * 4]) = w;
      }
    }
  };
  f32x4 acc[2][2] = {{z4, z4}, {z4, z4}};
  // a slice's products: the operands of k-step s + 1 are read from LDS while the four MFMAs of step s run -- a lone workgroup on its
  // CU (one wavefront per SIMD: grids of up to 256 tiles) otherwise waits out the LDS latency in front of every step (stamped: 1.15 us
  // per 32-deep slice against 0.45 us of MFMA issue; requesting the global loads two slices ahead instead changed nothing)
  auto readOps = [&](int buf, int s, float& a0, float& a1, float& b0, float& b1) {
    a0 = sA[buf][(wm + li) * MM_PA + 4 * s + lc]; a1 = sA[buf][(wm + 16 + li) * MM_PA + 4 * s + lc];
    if constexpr (TRANSW) { b0 = sB[buf][(wn + li) * MM_PA + 4 * s + lc]; b1 = sB[buf][(wn + 16 + li) * MM_PA + 4 * s + lc]; }
    else { b0 = sB[buf][(4 * s + lc) * MM_PB + wn + li]; b1 = sB[buf][(4 * s + lc) * MM_PB + wn + 16 + li]; }
  };
  auto mma = [&](int buf) {
    float a0, a1, b0, b1;
    readOps(buf, 0, a0, a1, b0, b1);
#pragma unroll
    for (int s = 0; s < MM_KS / 4; ++s) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (s + 1 < MM_KS / 4) readOps(buf, s + 1, na0, na1, nb0, nb1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  };
  // whole slices (K a multiple of the slice depth) of a tile inside the operands: row pointers formed once, a slice = one add per load,
  // nothing to mask (a lone workgroup on its CU issues a slice's address arithmetic and masks between its products, not beside them)
  const bool interior = K % MM_KS == 0 && m0 + 64 <= nRows && n0 + 64 <= P.N;
  const float* pa[Q]; const float* pb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    pa[q] = P.A + (size_t)(m0 + tid / PPR + RPP * q) * P.lda + (tid % PPR) * 4;
    if constexpr (TRANSW) pb[q] = P.B + (size_t)(n0 + tid / PPR + RPP * q) * P.ldb + (tid % PPR) * 4;
    else pb[q] = P.B + (size_t)((tid >> 4) + 16 * q) * P.ldb + n0 + (tid & 15) * 4;
  }
  const size_t stepB = TRANSW ? (size_t)MM_KS : (size_t)MM_KS * P.ldb;
  auto loadWhole = [&]() {      // (the next slice: the pointers advance)
#pragma unroll
    for (int q = 0; q < Q; ++q) { va[q] = *reinterpret_cast<const f32x4*>(pa[q]); vb[q] = *reinterpret_cast<const f32x4*>(pb[q]); pa[q] += MM_KS; pb[q] += stepB; }
  };
  auto storeWhole = [&](int buf) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      *reinterpret_cast<f32x4*>(&sA[buf][(tid / PPR + RPP * q) * MM_PA + (tid % PPR) * 4]) = va[q];
      if constexpr (TRANSW) *reinterpret_cast<f32x4*>(&sB[buf][(tid / PPR + RPP * q) * MM_PA + (tid % PPR) * 4]) = vb[q];
      else *reinterpret_cast<f32x4*>(&sB[buf][((tid >> 4) + 16 * q) * MM_PB + (tid & 15) * 4]) = vb[q];
    }
  };
  int buf = 0;
  if (interior) loadWhole(); else loadSlice(0);
  if (interior) {
    storeWhole(0);
    __syncthreads();
    MMSTAMP(2);
    for (int k0 = 0; k0 < K; k0 += MM_KS) {
      const bool more = k0 + MM_KS < K;
      if (more) loadWhole();
      mma(buf);
      if (more) storeWhole(buf ^ 1);      // (the other buffer: its readers passed the barrier of the previous slice)
      __syncthreads();
      buf ^= 1;
    }
  } else {
    storeSlice(0, 0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += MM_KS) {
      const bool more = k0 + MM_KS < K;
      if (more) loadSlice(k0 + MM_KS);
      mma(buf);
      if (more) storeSlice(buf ^ 1, k0 + MM_KS);
      __syncthreads();
      buf ^= 1;
    }
  }
  MMSTAMP(3);
  // ---- epilogue operands: every load of it requested here, in ONE round trip (inside the loop below each of the four row groups waited
  //      for its own: 4.2 of the 14 us of a lone tile's workgroup, stamped.  Requested in front of the slices they delay the first slice
  //      by what they save here -- the counter of outstanding loads is in order -- and cost 50 registers) ----
  const int ec = 4 * (tid & 15);
  // what later launches read of a hidden layer: f'(x) takes the pre-activation OR the output (actDiff), the layers above and the weight
  // gradients the block output (C3 behind a parametric residual, else the output)
  const bool keepX = !actDiffFromOutput(P.func), keepY = !keepX || P.C3 == nullptr;
  const float* actSrc = keepX ? P.actX : P.actY;
  float eb[4], ew[4], er[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = n0 + ec + e;
    eb[e] = (!TRANSW && n < P.N) ? P.bias[n] : 0.f;
    ew[e] = (n < P.resN && P.resW) ? P.resW[n] : 0.f;
    er[e] = (!TRANSW && n < P.resN && P.resB) ? P.resB[n] : 0.f;
  }
  f32x4 rinv[4], axv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + (tid >> 4) + 16 * q, n = n0 + ec;
    rinv[q] = z4; axv[q] = z4;
    if (m < nRows && n < P.N) {
      if ((TRANSW || P.C3) && n < P.resN && n + 3 < P.ldRes) rinv[q] = *reinterpret_cast<const f32x4*>(P.resIn + (size_t)m * P.ldRes + n);
      if constexpr (TRANSW) { if (n + 3 < P.ldAct) axv[q] = *reinterpret_cast<const f32x4*>(actSrc + (size_t)m * P.ldAct + n); }
    }
  }
  // ---- the tile through LDS (the slices are dead): sOut[64][64], columns rotated by 16 per group of four rows ----
  float* sOut = &sA[0][0];                       // 64 x 64 floats = 16 KB <= the two A buffers (18 KB)
  static_assert(2 * 64 * MM_PA >= 64 * 64, "output tile fits the A buffers");
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm + 16 * tm + 4 * lc + i, c = wn + 16 * tn + li;
        sOut[r * 64 + ((c + 16 * (r >> 2)) & 63)] = acc[tm][tn][i];
      }
  __syncthreads();
  MMSTAMP(4);
  // (the activation as a compile-time constant inside: one switch per workgroup instead of one per element)
  dispatchFunc<-1>(P.func, [&](auto F) {
  constexpr int FN = decltype(F)::value;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = (tid >> 4) + 16 * q, m = m0 + rr, n = n0 + ec;
    const f32x4 v = *reinterpret_cast<const f32x4*>(sOut + rr * 64 + ((ec + 16 * (rr >> 2)) & 63));
    if (m >= nRows || n >= P.N) continue;
    const size_t o = (size_t)m * P.ldc + n;
    const bool whole = n + 3 < P.N;
    const f32x4 rin = rinv[q];
    if constexpr (!TRANSW) {
      f32x4 x, y, r;
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = v[e] + eb[e]; y[e] = actEvalT<FN>(x[e]); r[e] = n + e < P.resN ? y[e] + (rin[e] * ew[e] + er[e]) : y[e]; }
      if (whole) {
        if (keepX) *reinterpret_cast<f32x4*>(P.C + o) = x;
        if (keepY) *reinterpret_cast<f32x4*>(P.C2 + o) = y;
        if (P.C3) *reinterpret_cast<f32x4*>(P.C3 + o) = r;
      } else for (int e = 0; e < 4; ++e) if (n + e < P.N) { if (keepX) P.C[o + e] = x[e]; if (keepY) P.C2[o + e] = y[e]; if (P.C3) P.C3[o + e] = r[e]; }
    } else {
      const f32x4 ax = axv[q];
      f32x4 dres, d;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dres[e] = n + e < P.resN ? v[e] + rin[e] * ew[e] : v[e]; d[e] = dres[e] * actDiffT<FN>(ax[e], ax[e]); }
      if (whole) { *reinterpret_cast<f32x4*>(P.C + o) = dres; *reinterpret_cast<f32x4*>(P.C2 + o) = d; }
      else for (int e = 0; e < 4; ++e) if (n + e < P.N) { P.C[o + e] = dres[e]; P.C2[o + e] = d[e]; }
    }
  }
  });
  MMSTAMP(6);
  MMEND();
}
bool big_mm_ok(const GemmProblem& P) {
  const bool al = (P.lda & 3) == 0 && (P.ldb & 3) == 0 && (P.ldc & 3) == 0 && (!P.resN || ((P.ldRes & 3) == 0 && P.ldRes >= ((P.resN + 3) & ~3))) &&
                  (P.flavor != GEMM_X || ((P.ldAct & 3) == 0 && P.ldAct >= ((P.N + 3) & ~3)));
  return (P.flavor == GEMM_F || P.flavor == GEMM_X) && al;
}
hipError_t launch_big_mm(const GemmProblem& P, const DevScalars* sc, int parity, hipStream_t s) {
  const int rowTiles = (P.M + 63) / 64, colTiles = (P.N + 63) / 64;
  const dim3 grid(8 * ((rowTiles + 7) / 8) * colTiles);
  // (measured at 256 and 512 tiles, K = 256: dX 18.6 -> 17.6 us with the deep slices, forward 17.7 -> 19.2 -- its three output arrays
  //  per tile are what it waits for, not its slices)
  const bool deep = P.flavor == GEMM_X && rowTiles * colTiles <= 512 && P.K >= 128;
  if (P.flavor == GEMM_X) { if (deep) hipLaunchKernelGGL((big_mm_kernel<true, 64>), grid, dim3(256), 0, s, P, sc, parity, rowTiles); else hipLaunchKernelGGL((big_mm_kernel<true, 32>), grid, dim3(256), 0, s, P, sc, parity, rowTiles); }
  else hipLaunchKernelGGL((big_mm_kernel<false, 32>), grid, dim3(256), 0, s, P, sc, parity, rowTiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int BD_PITCH = 80, BD_ROWS = 32;

__device__ __forceinline__ void bigDwBody(const GemmProblem& P, int blk, float (*sA)[BD_ROWS * BD_PITCH], float (*sD)[BD_ROWS * BD_PITCH]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int tilesN = (P.N + 63) / 64, tiles = ((P.M + 63) / 64) * tilesN, tile = blk % tiles, ks = blk / tiles;
  const int m0 = (tile / tilesN) * 64, n0 = (tile % tilesN) * 64;
  const int rBeg = ks * P.bigChunk, rEnd = min(P.K, rBeg + P.bigChunk);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // a slice = 32 rows x 64 columns of each operand: 512 float4 per operand, two per thread
  const int sr = tid >> 4, sc4 = (tid & 15) * 4;      // (+ 16 rows for the second one)
  f32x4 va[2], vd[2];
  const int caMax = max(0, min(P.lda, (P.M - 1 + 3) & ~3) - 4), cdMax = max(0, min(P.ldb, (P.N + 3) & ~3) - 4);
  // (loads with clamped addresses and nothing else; masks when the slice is stored -- as in big_mm_kernel: no wait for the data in
  //  front of the slice's products)
  auto loadSlice = [&](int r0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // (clamped to the PROBLEM's columns -- M - 1 inputs, N deltas, rounded up to a 16-byte unit --, not to the row pitch: the operand
      //  of a recurrent block starts inside its row (A = row + nIn), where pitch - 4 from that base lies behind the row's end)
      const int row = min(r0 + sr + 16 * q, rEnd - 1), ca = min(m0 + sc4, caMax), cd = min(n0 + sc4, cdMax);
      va[q] = *reinterpret_cast<const f32x4*>(P.A + (size_t)row * P.lda + ca);
      vd[q] = *reinterpret_cast<const f32x4*>(P.B + (size_t)row * P.ldb + cd);
    }
  };
  auto storeSlice = [&](int buf, int r0) {      // (r0: the slice the registers hold)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool in = r0 + sr + 16 * q < rEnd;
      const int ca = m0 + sc4, cd = n0 + sc4;
      f32x4 a = va[q], d = vd[q];
      // columns beyond the inputs: the ones column (bias row of the product) at M - 1, zeros behind it; deltas beyond N: zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!in || ca + e >= P.M) a[e] = 0.f; else if (ca + e == P.M - 1) a[e] = 1.f;
        if (!in || cd + e >= P.N) d[e] = 0.f;
      }
      *reinterpret_cast<f32x4*>(&sA[buf][(sr + 16 * q) * BD_PITCH + sc4]) = a;
      *reinterpret_cast<f32x4*>(&sD[buf][(sr + 16 * q) * BD_PITCH + sc4]) = d;
    }
  };
  f32x4 acc[2][2] = {{z4, z4}, {z4, z4}};
  loadSlice(rBeg); storeSlice(0, rBeg);
  __syncthreads();
  int buf = 0;
  for (int r0 = rBeg; r0 < rEnd; r0 += BD_ROWS) {
    const bool more = r0 + BD_ROWS < rEnd;
    if (more) loadSlice(r0 + BD_ROWS);
    {      // (the operands of step s + 1 are read while the MFMAs of step s run)
      const float* ra = &sA[buf][lc * BD_PITCH + wm + li];
      const float* rd = &sD[buf][lc * BD_PITCH + wn + li];
      float a0 = ra[0], a1 = ra[16], d0 = rd[0], d1 = rd[16];
#pragma unroll
      for (int s = 0; s < BD_ROWS / 4; ++s) {
        float na0 = 0.f, na1 = 0.f, nd0 = 0.f, nd1 = 0.f;
        if (s + 1 < BD_ROWS / 4) { const int o = 4 * (s + 1) * BD_PITCH; na0 = ra[o]; na1 = ra[o + 16]; nd0 = rd[o]; nd1 = rd[o + 16]; }
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, d0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, d1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, d0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, d1, acc[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; d0 = nd0; d1 = nd1;
      }
    }
    if (more) storeSlice(buf ^ 1, r0 + BD_ROWS);      // (the other buffer: its readers passed the barrier of the previous iteration)
    __syncthreads();
    buf ^= 1;
  }
  // partial tile -> part[ks][M][N] (splitk_reduce_kernel sums the chunks in order; row M - 1 is the bias)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn + 16 * tn + li;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + 16 * tm + 4 * lc + i;
        if (m < P.M && n < P.N) P.part[((size_t)ks * P.M + m) * P.N + n] = acc[tm][tn][i];
      }
    }
}
// column sums over the rows (RED_COL: C[c] = sum_rows A[row][c] (x B[row][c]): gradients of the parametric residual's weight and
// bias, of the policy's sigma parameters): a workgroup = 64 columns x one chunk of rows, 16 rows in flight per pass (256-byte row
// pieces), the 16 row lanes joined in LDS in order; partial sums to part[chunk][N] as for the products
__device__ __forceinline__ void bigRedBody(const GemmProblem& P, int blk, float* sm) {
  const int tilesN = (P.N + 63) / 64, n0 = (blk % tilesN) * 64, ks = blk / tilesN;
  const int rBeg = ks * P.bigChunk, rEnd = min(P.K, rBeg + P.bigChunk);
  const int tid = threadIdx.x, rl = tid >> 4, c = n0 + (tid & 15) * 4;
  const bool vec = (P.lda & 3) == 0 && (P.B == nullptr || (P.ldb & 3) == 0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  if (vec) {
    const bool in = c < P.lda && (P.B == nullptr || c < P.ldb);
    for (int r0 = rBeg + rl; r0 < rEnd; r0 += 16 * U) {
      f32x4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + 16 * u;
        va[u] = f32x4{0.f, 0.f, 0.f, 0.f}; vb[u] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (in && r < rEnd) {
          va[u] = *reinterpret_cast<const f32x4*>(P.A + (size_t)r * P.lda + c);
          if (P.B) vb[u] = *reinterpret_cast<const f32x4*>(P.B + (size_t)r * P.ldb + c);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += va[u] * vb[u];
    }
  } else {
    for (int r = rBeg + rl; r < rEnd; r += 16)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (c + e < P.N) acc[e] += P.A[(size_t)r * P.lda + c + e] * (P.B ? P.B[(size_t)r * P.ldb + c + e] : 1.f);
  }
  *reinterpret_cast<f32x4*>(sm + rl * 64 + (tid & 15) * 4) = acc;
  __syncthreads();
  if (tid < 64 && n0 + tid < P.N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q * 64 + tid];
    P.part[(size_t)ks * P.N + n0 + tid] = t;
  }
}
// every weight gradient of a large-batch step in ONE launch: workgroup -> (problem of the list, tile or column block, row chunk)
__global__ __launch_bounds__(256, 2) void big_dw_kernel(const GemmProblem* __restrict__ probs, BigDwList L) {
  __shared__ __attribute__((aligned(16))) float sA[2][BD_ROWS * BD_PITCH], sD[2][BD_ROWS * BD_PITCH];
  int k = 0, blk = (int)blockIdx.x;
#pragma unroll
  for (int q = 1; q < BIG_DW_MAX; ++q) if (q < L.n && blk >= L.start[q]) k = q;
  blk -= L.start[k];
  const GemmProblem& P = probs[L.idx[k]];
  if (P.flavor == RED_COL) bigRedBody(P, blk, &sA[0][0]); else bigDwBody(P, blk, sA, sD);
}
// rows per chunk: about 640 workgroups per product (tiles x chunks), 512 per column-sum problem; whole 32-row slices, 64 rows at least
int big_dw_chunk_rows(const GemmProblem& P) {
  const bool red = P.flavor == RED_COL;
  const int tiles = red ? (P.N + 63) / 64 : ((P.M + 63) / 64) * ((P.N + 63) / 64);
  const int chunks = std::max(2, (red ? 512 : 640) / tiles);
  int per = (P.K + chunks - 1) / chunks;
  per = std::max(64, (per + BD_ROWS - 1) / BD_ROWS * BD_ROWS);
  return per;
}
int big_dw_blocks(const GemmProblem& P) {
  const int tiles = P.flavor == RED_COL ? (P.N + 63) / 64 : ((P.M + 63) / 64) * ((P.N + 63) / 64);
  return tiles * P.nSplit;
}
bool big_dw_ok(const GemmProblem& P) {
  if (P.flavor == RED_COL) return true;      // (any pitch: rows that are not 16-byte aligned take scalar loads)
  return P.flavor == GEMM_W && (P.lda & 3) == 0 && (P.ldb & 3) == 0;
}
hipError_t launch_big_dw(const GemmProblem* dProbs, const BigDwList& L, hipStream_t s) {
  if (L.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(big_dw_kernel, dim3(L.start[L.n]), dim3(256), 0, s, dProbs, L);
  return hipGetLastError();
}

}  // namespace hl
