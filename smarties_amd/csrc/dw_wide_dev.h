// smarties_amd/csrc/dw_wide_dev.h -- device body of the wide weight-gradient launch (gemm16.hip: dw_wide_kernel), shared with the
// convolutional step's merged weight-gradient launch (conv.hip: conv_dw_dense_kernel), where the dense layers' tiles run beside the
// filter-gradient workgroups (Layer::backward dW, Layers.h:164-187; Adam::step, Optimizer.cpp:61-108).
#pragma once
#include "gemm_tile.h"

namespace hl {

// the reduction of one wavefront: rows [r0, rEnd) of A (columns m0 ..) and B (columns n0 ..), UN steps of four rows per batch of loads
template <int UN>
__device__ __forceinline__ void dwwRows(const GemmProblem& P, const float* pA, const float* pB, bool aOne, bool aOk, bool bOk, int r0, int rEnd, int lc,
                                        f32x4& acc0, f32x4& acc1) {
  for (int rb = r0; rb < rEnd; rb += 4 * UN) {
    float av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {      // (clamped rows: no predicated loads; what they fetch is masked below)
      const unsigned r = (unsigned)min(rb + 4 * u + lc, P.K - 1);
      av[u] = pA[r * (unsigned)P.lda]; bv[u] = pB[r * (unsigned)P.ldb];      // (32-bit element offsets from the uniform bases)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; u += 2) {
      const bool in0 = rb + 4 * u + lc < rEnd, in1 = rb + 4 * (u + 1) + lc < rEnd;
      const float a0 = in0 ? (aOne ? 1.f : (aOk ? av[u] : 0.f)) : 0.f, b0 = (in0 && bOk) ? bv[u] : 0.f;
      const float a1 = in1 ? (aOne ? 1.f : (aOk ? av[u + 1] : 0.f)) : 0.f, b1 = (in1 && bOk) ? bv[u + 1] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
  }
}
constexpr int DWW_NT = 256, DWW_NW = DWW_NT / 64, DWW_UN = 34;      // (four wavefronts: three workgroups per CU at the kernel's ~150 VGPRs -- 66 tiles x 4 quarters of the LSTM shape in one round; the riders are written for 256 threads)
// workgroup `bid` of the launch's tile workgroups (riders not counted); smem: DWW_LDS bytes, 16-byte aligned
constexpr int DWW_LDS = DWW_NW * 256 * 4 + 16;
// MAXUN: the longest batch of row steps instantiated (DWW_UN: any row count; 16: at most 256 rows per workgroup -- fewer registers)
template <int MAXUN = DWW_UN>
__device__ __forceinline__ void dwWideBody(const GemmProblem* __restrict__ probs, int nProbs, int nTiles, int nq, float* __restrict__ part,
                                           unsigned* __restrict__ ctr, const DevScalars* __restrict__ sc, const AdamHyper& hyp, int bid, unsigned char* smem) {
  // the nq workgroups of a tile sit 8 block indices apart (same XCD where block b runs on XCD b % 8: the hand-off then stays in one L2)
  const int grp = bid / (8 * nq), rem = bid - grp * (8 * nq), qk = rem >> 3, gt = grp * 8 + (rem & 7);
  if (gt >= nTiles) return;
  int p = 0;
  for (int i = 1; i < nProbs; ++i) if (gt >= probs[i].tileStart) p = i;
  const GemmProblem P = probs[p];
  const int tile = gt - P.tileStart;
  float* red = reinterpret_cast<float*>(smem);
  unsigned* sArr = reinterpret_cast<unsigned*>(smem + DWW_NW * 256 * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // rows of this workgroup's share: a multiple of four
  const int KQ = (((P.K + nq - 1) / nq) + 3) & ~3, q0 = qk * KQ, qEnd = min(P.K, q0 + KQ);
  float* myPart = part + ((size_t)gt * DW_WIDE_Q) * 256;
  if (P.flavor == RED_COL) {      // out[j] = sum_m A[m][j] (B ? B[m][j] : 1): 16 row partitions per share, joined in partition, then share order
    constexpr int NP = DWW_NT / 16, RU = 34;
    const int jj = tid & 15, prt = tid >> 4, j = tile * 16 + jj;
    float acc = 0.f;
    if (j < P.N) {
      if (MAXUN < DWW_UN || qEnd - q0 <= NP * 8) {      // (minibatch rows: one short batch)
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int mr = q0 + prt + NP * u;
          av[u] = mr < qEnd ? P.A[(size_t)mr * P.lda + j] : 0.f;
          bv[u] = (P.B && mr < qEnd) ? P.B[(size_t)mr * P.ldb + j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += av[u] * bv[u];
      } else
      for (int mb = q0 + prt; mb < qEnd; mb += NP * RU) {
        float av[RU], bv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int mr = mb + NP * u;
          av[u] = mr < qEnd ? P.A[(size_t)mr * P.lda + j] : 0.f;
          bv[u] = (P.B && mr < qEnd) ? P.B[(size_t)mr * P.ldb + j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) acc += av[u] * bv[u];
      }
    }
    red[prt * 16 + jj] = acc;
    __syncthreads();
    float g = 0.f;
    if (tid < 16) for (int q = 0; q < NP; ++q) g += red[q * 16 + tid];
    if (nq > 1) {
      if (tid < 16) __hip_atomic_store(myPart + qk * 256 + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) sArr[0] = __hip_atomic_fetch_add(ctr + gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (sArr[0] % (unsigned)nq != (unsigned)(nq - 1)) return;
      g = 0.f;
      if (tid < 16) for (int q = 0; q < nq; ++q) g += __hip_atomic_load(myPart + q * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 16 && j < P.N) {
      P.C[j] = g;
      if (P.adam) { AdamCoef c; c.eta = sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac; adamApply(c, g, P.adW, P.adM1, P.adM2, j); }
    }
    return;
  }
  const int tm = tile / P.tilesN, tn = tile - tm * P.tilesN;
  const int m0 = tm * 16, n0 = tn * 16;
  const int li = lane & 15, lc = lane >> 4;
  // epilogue operands of this thread's element, requested with the tile's (used by the last workgroup to arrive)
  const int m = m0 + ((tid >> 4) & 15), n = n0 + (tid & 15);
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
  // rows of this wavefront: a multiple of four
  const int RW = (((KQ + DWW_NW - 1) / DWW_NW) + 3) & ~3;
  const int r0 = q0 + wave * RW, rEnd = min(qEnd, r0 + RW);
  const int ma = m0 + li, nb = n0 + li;
  const bool aOne = ma == P.M - 1, aOk = ma < P.M - 1 && ma < P.lda, bOk = nb < P.N;
  const float* pA = P.A + (aOk ? ma : 0);
  const float* pB = P.B + (bOk ? nb : 0);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (RW <= 32) dwwRows<8>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);      // (minibatch rows: eight steps per wavefront at 128)
  else if (RW <= 64) dwwRows<16>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);
  else if (MAXUN >= DWW_UN) dwwRows<DWW_UN>(P, pA, pB, aOne, aOk, bOk, r0, rEnd, lc, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  if (nq > 1) {
    __hip_atomic_store(myPart + qk * 256 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): the share's tile is at the coherence point
    __syncthreads();
    if (tid == 0) sArr[0] = __hip_atomic_fetch_add(ctr + gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sArr[0] % (unsigned)nq != (unsigned)(nq - 1)) return;      // not the last of this step's nq
    v = 0.f;
    for (int q = 0; q < nq; ++q) v += __hip_atomic_load(myPart + q * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!outOk) return;
  {      // (one store sequence on selected pointer values: gemm_tile.h)
    const bool isW = m < P.M - 1;
    const size_t i = isW ? (size_t)m * P.ldc + n : (size_t)n;
    pickPtrW(isW, P.C, P.biasOut)[i] = v;
    if (P.adam) { adamStep(ac, v, e0, e1, e2); pickPtrW(isW, P.adW, P.adbW)[i] = e0; pickPtrW(isW, P.adM1, P.adbM1)[i] = e1; pickPtrW(isW, P.adM2, P.adbM2)[i] = e2; }
  }
}

}  // namespace hl
