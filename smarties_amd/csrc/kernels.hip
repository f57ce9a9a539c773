// smarties_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels of the
// V-RACER / ReF-ER learner update.  64-wide wavefronts, fp32 MFMA (v_mfma_f32_16x16x4_f32,
// bit-exact fp32 fma chains) for the MLP contractions, LDS-staged tiles, fp64 head math.
//
// One gradient step = sample_kernel -> gemm16 (one launch per hidden layer, forward) ->
// head_kernel -> gemm16 (dX, one launch per hidden layer but the first) -> gemm16 (all dW /
// bias / residual-parameter gradients in ONE multi-problem launch) -> [RCCL all-reduce] ->
// adam_kernel -> post_kernel.  Reference call sites are cited per kernel.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include "kernels.h"

namespace hl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float actEval(int f, float in) {   // Network/Layers/Functions.h
  switch (f) {
    case HL_FUNC_TANH:
      if (in > 0) { const float e = expf(-2 * in); return (1 - e) / (1 + e); }
      else        { const float e = expf( 2 * in); return (e - 1) / (1 + e); }
    case HL_FUNC_SOFTSIGN: return in / (1 + fabsf(in));
    case HL_FUNC_RELU: return in > 0 ? in : 0.f;
    default: return in;
  }
}
__device__ __forceinline__ float actDiff(int f, float in, float out) {
  switch (f) {
    case HL_FUNC_TANH: return 1 - out * out;
    case HL_FUNC_SOFTSIGN: { const float d = 1 + fabsf(in); return 1 / (d * d); }
    case HL_FUNC_RELU: return in > 0 ? 1.f : 0.f;
    default: return 1.f;
  }
}
// Far-policy steps an episode contributes to ReplayStats::nFarPolicySteps.  The reference adds
// the float Nsteps*fracFarPolSteps to an integer counter with a truncation after every add
// (MemoryProcessing.cpp:227): a product a few ulps below an integer still lands on that integer
// because the float add rounds, a genuinely fractional product (N/(N-1) after a recompute) is
// truncated.  floor(x + 1e-3) reproduces both cases independently of the summation order.
__device__ __forceinline__ long long farSteps(float Nsteps, float fracFar) {
  return (long long)floorf(Nsteps * fracFar + 1e-3f);
}
__device__ __forceinline__ double waveSum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float waveSumF(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------
// gemm16_kernel: C(16x16 tile per workgroup) with the reduction split over the 4 waves.
// Operand tiles are staged through LDS with coalesced global reads; fragments are read with
// conflict-free ds_read_b32 (ROWS tile: leading dim 258 == 2 mod 32 banks; COLS tile: 16).
// Replaces BaseLayer::forward (Layer_Base.h:64-95), ParametricResidualLayer::forward/backward
// (Layers.h:347-393) and Layer::backward (Layers.h:123-188) over the whole minibatch.
// ---------------------------------------------------------------------------
#define KC 256
#define LDR 258

__device__ __forceinline__ void redcol_tile(const GemmProblem& P, int tile, float* red) {
  // out[j] = sum_m A[m][j] * (B ? B[m][j] : 1),  64 columns per workgroup, 4 row-partitions
  const int tid = threadIdx.x, jj = tid & 63, part = tid >> 6;
  const int j = tile * 64 + jj;
  float acc = 0.f;
  if (j < P.N) {
    for (int m = part; m < P.K; m += 4) {
      const float a = P.A[(size_t)m * P.lda + j];
      acc += P.B ? a * P.B[(size_t)m * P.ldb + j] : a;
    }
  }
  red[part * 64 + jj] = acc;
  __syncthreads();
  if (part == 0 && j < P.N) P.C[j] = (red[jj] + red[64 + jj]) + (red[128 + jj] + red[192 + jj]);
}

__global__ __launch_bounds__(256) void gemm16_kernel(const GemmProblem* __restrict__ probs, int nProbs,
                                                     const DevScalars* __restrict__ sc) {
  __shared__ float sA[16 * LDR];
  __shared__ float sB[16 * LDR];
  __shared__ float red[4 * 256];
  const int bid = blockIdx.x;
  int p = 0;
  for (int i = 1; i < nProbs; ++i) if (bid >= probs[i].tileStart) p = i;
  const GemmProblem P = probs[p];
  const int tile = bid - P.tileStart;
  if (P.flavor == RED_COL) { redcol_tile(P, tile, red); return; }

  const int tm = tile / P.tilesN, tn = tile - tm * P.tilesN;
  const int m0 = tm * 16, n0 = tn * 16;
  const int Mvalid = P.dynRows ? sc->nRows : P.M;
  if (m0 >= Mvalid) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lc = lane >> 4;
  const bool aRows = (P.flavor != GEMM_W);   // A tile is 16 rows x k  (else k x 16)
  const bool bRows = (P.flavor == GEMM_X);   // B tile is 16 rows x k  (else k x 16)

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < P.K; kb += KC) {
    const int kc = min(KC, P.K - kb);
    int kw = (kc + 3) / 4; kw = (kw + 7) & ~7;        // k handled by each wave, multiple of 8
    const int kcp = 4 * kw;                           // <= 256
    // ---- stage A ----
    if (aRows) {
      for (int e = tid; e < 16 * kcp; e += 256) {
        const int r = e / kcp, c = e - r * kcp;
        float v = 0.f;
        if (m0 + r < Mvalid && c < kc) v = P.A[(size_t)(m0 + r) * P.lda + kb + c];
        sA[r * LDR + c] = v;
      }
    } else {  // GEMM_W: A^T, source acts[k][m0+i]; row M-1 of the product is the ones row (bias grad)
      for (int e = tid; e < 16 * kcp; e += 256) {
        const int k = e >> 4, i = e & 15;
        float v = 0.f;
        if (k < kc) {
          const int row = m0 + i;
          if (row < P.M - 1) v = P.A[(size_t)(kb + k) * P.lda + row];
          else if (row == P.M - 1) v = 1.f;
        }
        sA[e] = v;
      }
    }
    // ---- stage B ----
    if (bRows) {   // GEMM_X: weights rows n0.., columns = reduction
      for (int e = tid; e < 16 * kcp; e += 256) {
        const int r = e / kcp, c = e - r * kcp;
        float v = 0.f;
        if (n0 + r < P.N && c < kc) v = P.B[(size_t)(n0 + r) * P.ldb + kb + c];
        sB[r * LDR + c] = v;
      }
    } else {
      for (int e = tid; e < 16 * kcp; e += 256) {
        const int k = e >> 4, j = e & 15;
        float v = 0.f;
        if (k < kc && n0 + j < P.N) v = P.B[(size_t)(kb + k) * P.ldb + n0 + j];
        sB[e] = v;
      }
    }
    __syncthreads();
    const int k0 = wave * kw;
#pragma unroll 4
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
  // ---- cross-wave reduction of the 4 partial tiles ----
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
  if (m >= Mvalid || n >= P.N) return;

  if (P.epi == EPI_FWD) {
    const float x = v + P.bias[n];
    P.C[(size_t)m * P.ldc + n] = x;
    const float y = actEval(P.func, x);
    P.C2[(size_t)m * P.ldc + n] = y;
    if (P.C3) {
      float r = y;
      if (n < P.resN) r += P.resIn[(size_t)m * P.ldRes + n] * P.resW[n] + P.resB[n];
      P.C3[(size_t)m * P.ldc + n] = r;
    }
  } else if (P.epi == EPI_DX) {
    float dres = v;
    if (n < P.resN) dres += P.resIn[(size_t)m * P.ldRes + n] * P.resW[n];
    P.C[(size_t)m * P.ldc + n] = dres;
    P.C2[(size_t)m * P.ldc + n] =
        dres * actDiff(P.func, P.actX[(size_t)m * P.ldAct + n], P.actY[(size_t)m * P.ldAct + n]);
  } else if (P.epi == EPI_DW) {
    if (m < P.M - 1) P.C[(size_t)m * P.ldc + n] = v;
    else P.biasOut[n] = v;
  } else {
    P.C[(size_t)m * P.ldc + n] = v;
  }
}

hipError_t launch_gemm(const GemmProblem* dProbs, int nProbs, int nBlocks, const DevScalars* sc, hipStream_t s) {
  if (nBlocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(gemm16_kernel, dim3(nBlocks), dim3(256), 0, s, dProbs, nProbs, sc);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sample_kernel (one workgroup): device-side restatement of
//   Sample_uniform::sample + Sampling::IDtoSeqStep (ReplayMemory/Sampling.cpp:26-47,82-96)
//   over std::mt19937 generators[0] with libstdc++'s uniform_int_distribution (Lemire),
//   the minibatch gather of MemoryBuffer::sampleMinibatch (MemoryBuffer.cpp:413-429) and the
//   per-Adam-step generator draw (Network/Optimizer.cpp:139).
// The Mersenne-twister regeneration is done cooperatively by the workgroup in LDS.
// ---------------------------------------------------------------------------
#define SMAXB 2048

__device__ __forceinline__ unsigned mtTemper(unsigned z) {
  z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
  return z;
}
__device__ __forceinline__ unsigned mtF(unsigned a, unsigned b) {
  const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// all threads of the block call this; xo/xn are LDS arrays of 624 words
__device__ void mtTwist(unsigned* x, unsigned* xo) {
  const int tid = threadIdx.x;
  for (int k = tid; k < 624; k += blockDim.x) xo[k] = x[k];
  __syncthreads();
  for (int k = tid; k < 227; k += blockDim.x) x[k] = xo[k + 397] ^ mtF(xo[k], xo[k + 1]);
  __syncthreads();
  for (int k = 227 + tid; k < 454; k += blockDim.x) x[k] = x[k - 227] ^ mtF(xo[k], xo[k + 1]);
  __syncthreads();
  for (int k = 454 + tid; k < 623; k += blockDim.x) x[k] = x[k - 227] ^ mtF(xo[k], xo[k + 1]);
  __syncthreads();
  if (tid == 0) x[623] = x[396] ^ mtF(xo[623], x[0]);
  __syncthreads();
}
// append n raw tempered words to raw[0..n) (all threads call; *pPos is in LDS)
__device__ void mtDraw(unsigned* x, unsigned* xo, int* pPos, unsigned* raw, int n) {
  int done = 0;
  while (done < n) {
    int pos = *pPos;
    __syncthreads();
    if (pos >= 624) { mtTwist(x, xo); pos = 0; }
    const int take = min(n - done, 624 - pos);
    for (int i = threadIdx.x; i < take; i += blockDim.x) raw[done + i] = mtTemper(x[pos + i]);
    __syncthreads();
    if (threadIdx.x == 0) *pPos = pos + take;
    __syncthreads();
    done += take;
  }
}
// exclusive scan of flags[0..n) into out[0..n), returns total (all threads call)
__device__ int blockScan(const int* flags, int* out, int n, int* tmp) {
  // chunked: thread t owns elements [t*per, (t+1)*per)
  const int T = blockDim.x, per = (n + T - 1) / T, t = threadIdx.x;
  int s = 0;
  for (int i = t * per; i < min(n, (t + 1) * per); ++i) s += flags[i];
  tmp[t] = s;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {
    int v = (t >= off) ? tmp[t - off] : 0;
    __syncthreads();
    tmp[t] += v;
    __syncthreads();
  }
  int base = (t == 0) ? 0 : tmp[t - 1];
  const int total = tmp[T - 1];
  for (int i = t * per; i < min(n, (t + 1) * per); ++i) { out[i] = base; base += flags[i]; }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  __shared__ unsigned x[624], xo[624];
  __shared__ unsigned raw[SMAXB];
  __shared__ unsigned long long vals[SMAXB];
  __shared__ int flags[SMAXB], scanv[SMAXB], tmp[256];
  __shared__ int sPos, sHave;
  const int tid = threadIdx.x, B = a.B;
  DevScalars* sc = a.sc;
  for (int k = tid; k < 624; k += 256) x[k] = sc->rng[k];
  if (tid == 0) sPos = (int)sc->rngPos;
  __syncthreads();
  const unsigned long long nData = (unsigned long long)sc->nTransitions;

  if (a.flatGiven) {
    for (int i = tid; i < B; i += 256) vals[i] = (unsigned long long)a.flatGiven[i];
    __syncthreads();
  } else {
    const unsigned range = (unsigned)nData;
    const unsigned threshold = (0u - range) % range;
    int Bp = 1; while (Bp < B) Bp <<= 1;
    int have = 0;                       // vals[0..have) = sorted unique prefix
    while (have < B) {
      // ---- draw B-have accepted values (Lemire rejection, words consumed in order) ----
      int filled = have;
      while (filled < B) {
        const int need = B - filled;
        mtDraw(x, xo, &sPos, raw, need);
        for (int i = tid; i < need; i += 256) {
          const unsigned long long prod = (unsigned long long)raw[i] * (unsigned long long)range;
          flags[i] = ((unsigned)prod >= threshold) ? 1 : 0;
        }
        __syncthreads();
        const int acc = blockScan(flags, scanv, need, tmp);
        for (int i = tid; i < need; i += 256)
          if (flags[i]) vals[filled + scanv[i]] = ((unsigned long long)raw[i] * (unsigned long long)range) >> 32;
        __syncthreads();
        filled += acc;
      }
      // ---- bitonic sort of vals[0..B) padded to Bp ----
      for (int i = B + tid; i < Bp; i += 256) vals[i] = ~0ull;
      __syncthreads();
      for (int k = 2; k <= Bp; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < Bp; i += 256) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const unsigned long long vi = vals[i], vj = vals[ixj];
              const bool up = ((i & k) == 0);
              if ((vi > vj) == up) { vals[i] = vj; vals[ixj] = vi; }
            }
          }
          __syncthreads();
        }
      // ---- std::unique ----
      for (int i = tid; i < B; i += 256) flags[i] = (i == 0 || vals[i] != vals[i - 1]) ? 1 : 0;
      __syncthreads();
      const int nu = blockScan(flags, scanv, B, tmp);
      unsigned long long keep[SMAXB / 256];
      for (int i = tid, q = 0; i < B; i += 256, ++q) keep[q] = vals[i];
      __syncthreads();
      for (int i = tid, q = 0; i < B; i += 256, ++q) if (flags[i]) vals[scanv[i]] = keep[q];
      __syncthreads();
      have = nu;
    }
  }
  // ---- the generator draws of AdamOptimizer::apply_update (one per reference thread) ----
  if (a.adamDraws > 0) mtDraw(x, xo, &sPos, raw, a.adamDraws);

  // ---- IDtoSeqStep: flat index -> (episode position, step) by binary search on the prefix ----
  const int nEp = (int)sc->nEpisodes;
  for (int b = tid; b < B; b += 256) {
    const long long f = (long long)vals[b];
    int lo = 0, hi = nEp;             // largest k with prefix[k] <= f
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.rp.posPrefix[mid] <= f) lo = mid; else hi = mid; }
    const int e = a.rp.posEid[lo];
    const int t = (int)(f - a.rp.posPrefix[lo]);
    a.bt.flat[b] = f; a.bt.pos[b] = lo; a.bt.eid[b] = e; a.bt.t[b] = t; a.bt.tag[b] = a.rp.epTag[e];
    a.bt.slot[b] = a.rp.epOff[e] + t;
    // Episode::isTruncated(t+1) (Episode.h:158-161)
    flags[b] = (t + 2 == a.rp.epN[e] && !a.rp.epTerm[e]) ? 1 : 0;
  }
  __syncthreads();
  const int nNext = blockScan(flags, scanv, B, tmp);
  for (int b = tid; b < B; b += 256) {
    if (flags[b]) { a.bt.nextOf[b] = B + scanv[b]; a.bt.nextSrc[scanv[b]] = b; }
    else a.bt.nextOf[b] = -1;
  }
  // ---- gather: Episode::standardizedState (Episode.h:172-183) for s_t and truncated s_{t+1} ----
  const int dS = a.dS;
  __syncthreads();
  for (int e = tid; e < B * dS; e += 256) {
    const int b = e / dS, i = e - b * dS;
    const long long slot = a.bt.slot[b];
    a.X0[(size_t)b * a.ldX0 + i] = (a.rp.S[(size_t)slot * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
    const int nr = a.bt.nextOf[b];
    if (nr >= 0)
      a.X0[(size_t)nr * a.ldX0 + i] = (a.rp.S[(size_t)(slot + 1) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
  }
  // ---- write back generator state ----
  for (int k = tid; k < 624; k += 256) sc->rng[k] = x[k];
  if (tid == 0) { sc->rngPos = (unsigned)sPos; sc->nNext = nNext; sc->nRows = B + nNext; }
}

hipError_t launch_sample(const SampleArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// head_kernel: one wavefront per minibatch row.
//   output InnerProduct layer (Linear) + ParamLayer (Layers.h:510-520),
//   RACER::Train for VRACER (Learners/RACER_train.cpp:14-67) in fp64 with
//   Continuous_policy (Math/Continuous_policy.h:68-378,569-738),
//   write-backs MiniBatch::setMseDklImpw / setValues (MiniBatch.h:161-175),
//   backward of the output layer into the last hidden block (Layers.h:123-160).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double scaleNet2V(double x) {   // RACER_common.cpp:23-27
  return x > 0 ? 100 * (x + 51) - 100 * sqrt(2601 + 100 * x) : 100 * (x - 51) + 100 * sqrt(2601 - 100 * x);
}
__device__ __forceinline__ double scaleVdiff(double x) {   // RACER_common.cpp:28-32
  return x > 0 ? 100 - 5000 / sqrt(2601 + 100 * x) : 100 - 5000 / sqrt(2601 - 100 * x);
}

#define HEAD_MAXOUT 136
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  __shared__ double sO[4][HEAD_MAXOUT];
  __shared__ float sDelta[4][72];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const DevScalars* sc = a.sc;
  if (row >= sc->nRows) return;
  const int B = a.B, dA = a.dA, nDense = a.nDense, H = a.H;
  const bool isNext = row >= B;
  const int b = isNext ? a.bt.nextSrc[row - B] : row;
  const float* y = a.Yin + (size_t)row * a.ldY;
  const float* Wo = a.params + a.indWo;
  // ---- output dense layer: O[o] = b[o] + sum_k y[k] W[k][o] ----
  const int nChunkOut = isNext ? 1 : nDense;   // a next-state row only needs V = O[0]
  for (int o0 = 0; o0 < nChunkOut; o0 += 8) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = lane; k < H; k += 64) {
      const float yk = y[k];
      const float* w = Wo + (size_t)k * a.ldWo + o0;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (o0 + q < nDense) p[q] += yk * w[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = waveSumF(p[q]);
    if (lane < 8 && o0 + lane < nDense) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q == lane) v = p[q];
      sO[wave][o0 + lane] = (double)(v + a.params[a.indBo + o0 + lane]);
    }
  }
  if (lane < dA) sO[wave][nDense + lane] = (double)a.params[a.indBp + lane];   // ParamLayer, Linear
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  const long long slot = a.bt.slot[b];
  if (isNext) {   // RACER_train.cpp:23-27: V(s_{t+1}) of a truncated episode end
    if (lane == 0) {
      const float Vn = (float)scaleNet2V(sO[wave][0]);
      a.bt.oldNextV[b] = a.rp.V[slot + 1]; a.bt.oldNextADV[b] = a.rp.ADV[slot + 1];
      a.rp.V[slot + 1] = Vn; a.rp.ADV[slot + 1] = 0.f; a.bt.nextV[b] = Vn;
      a.bt.O[(size_t)row * a.nOut] = sO[wave][0];
    }
    return;
  }

  // ---- policy terms, one action component per lane ----
  const double beta = sc->beta, Cmax = sc->Cmax, Cinv = sc->Cinv;
  const double MAXM = 8.31776613503286, LOG2PI_2 = 9.1893853320467266954096885456237942e-01;
  double lw = 0, kl = 0, mean = 0, stdev = 1, invStd = 1, dPos = 0, act = 0, bMean = 0, bStd = 1;
  bool bnd = false;
  if (lane < dA) {
    const int i = lane;
    bnd = a.bounded[i] != 0;
    mean = sO[wave][1 + i];
    const double pp = sO[wave][nDense + i];
    stdev = (pp + sqrt(1 + pp * pp)) / 2; invStd = 1 / stdev; dPos = (1 + pp / sqrt(1 + pp * pp)) / 2;
    act = a.rp.A[(size_t)slot * dA + i];
    bMean = a.rp.MU[(size_t)slot * 2 * dA + i]; bStd = a.rp.MU[(size_t)slot * 2 * dA + dA + i];
    const double bInv = 1 / bStd;
    double lpPi, lpMu;
    if (bnd) {
      const double m = mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean);
      const double sq = tanh(act), J = fmax(1 - sq * sq, (double)FLT_MIN);
      const double u1 = (act - m) * invStd, u2 = (act - bMean) * bInv;
      lpPi = -(u1 * u1) / 2 + log(invStd / J) - LOG2PI_2;
      lpMu = -(u2 * u2) / 2 + log(bInv / J) - LOG2PI_2;
    } else {
      const double u1 = (act - mean) * invStd, u2 = (act - bMean) * bInv;
      lpPi = -(u1 * u1) / 2 + log(invStd) - LOG2PI_2;
      lpMu = -(u2 * u2) / 2 + log(bInv) - LOG2PI_2;
    }
    lw = lpPi - lpMu;
    const double q = stdev / bStd, CmuCpi = q * q, dm = (mean - bMean) / bStd;
    kl = (CmuCpi - 1 + dm * dm - log(CmuCpi)) / 2;
  }
  const double logW = waveSum(lw), DKL = waveSum(kl);
  const double RHO = exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
  const float Wf = (float)RHO, Cf = (float)Cmax, iCf = (float)Cinv;
  const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);          // Episode.h:28-33 (Fval)
  const double O0 = sO[wave][0];
  const double V = scaleNet2V(O0);
  const double Qret = (double)a.rp.RET[slot];
  const double A_RET = Qret - V, dQ = A_RET;                       // Zero_advantage
  const double Ver = fmin(1.0, RHO) * dQ;
  const double g0 = far ? 0.0 : Ver * beta * scaleVdiff(O0);
  const double coef = A_RET * fmin(Cmax, RHO);
  double gM = 0, gS = 0;
  if (lane < dA) {
    const double dMean = mean - bMean, invVarMu = 1 / (bStd * bStd);
    const double penalM = -1 * (dMean * invVarMu);
    const double penalS = dPos * -1 * ((invVarMu - invStd * invStd) * stdev);
    double polM = 0, polS = 0;
    if (!far) {
      if (bnd) {
        const double dLogPdMean = (act - mean) * invStd * invStd;
        const double m = mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean);
        const double u = (act - m) * invStd;
        polS = dPos * coef * ((u * u - 1) * invStd);
        if (mean >= MAXM && coef * dLogPdMean > 0) polM = 0;
        else if (mean <= -MAXM && coef * dLogPdMean < 0) polM = 0;
        else polM = coef * dLogPdMean;
      } else {
        const double u = (act - mean) * invStd;
        polM = coef * (u * invStd);
        polS = dPos * coef * ((u * u - 1) * invStd);
      }
    }
    gM = beta * polM + (1 - beta) * penalM;
    gS = beta * polS + (1 - beta) * penalS;
    // Activation::addOutputDelta: nnReal += Real (Activation.h:108-117)
    sDelta[wave][1 + lane] = (float)gM;
    a.bt.gParam[(size_t)b * dA + lane] = (float)gS;
    a.bt.G[(size_t)b * a.nOut + 1 + lane] = (double)(float)gM;
    a.bt.G[(size_t)b * a.nOut + nDense + lane] = (double)(float)gS;
  }
  if (lane == 0) {
    sDelta[wave][0] = (float)g0;
    a.bt.G[(size_t)b * a.nOut] = (double)(float)g0;
    a.bt.rho[b] = RHO; a.bt.dkl[b] = DKL; a.bt.far[b] = far ? 1 : 0;
    // write-backs (Fval casts, MiniBatch.h:161-175); old values kept for the aggregate updates
    const float E = (float)dQ, D = (float)DKL, Wn = (float)RHO, Vf = (float)V;
    a.bt.oldDQ[b] = a.rp.DQ[slot]; a.bt.oldDKL[b] = a.rp.DKL[slot]; a.bt.oldW[b] = a.rp.IMPW[slot];
    a.bt.oldV[b] = a.rp.V[slot]; a.bt.oldADV[b] = a.rp.ADV[slot];
    a.bt.newDQ[b] = E; a.bt.newDKL[b] = D; a.bt.newW[b] = Wn; a.bt.newV[b] = Vf;
    a.rp.DQ[slot] = E; a.rp.DKL[slot] = D; a.rp.IMPW[slot] = Wn; a.rp.V[slot] = Vf; a.rp.ADV[slot] = 0.f;
    a.bt.dq[b] = (double)E;
  }
  for (int o = lane; o < a.nOut; o += 64) a.bt.O[(size_t)row * a.nOut + o] = sO[wave][o];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // ---- deltas of the output layer and back-propagation into the last hidden block ----
  for (int o = lane; o < nDense; o += 64) a.dOut[(size_t)b * a.ldDo + o] = sDelta[wave][o];
  for (int k = lane; k < H; k += 64) {
    const float* w = Wo + (size_t)k * a.ldWo;
    float s = 0.f;
    for (int o = 0; o < nDense; ++o) s += w[o] * sDelta[wave][o];
    a.Dres[(size_t)b * a.ldD + k] = s;
    a.D[(size_t)b * a.ldD + k] =
        s * actDiff(a.func, a.Xlast[(size_t)row * a.ldD + k], a.Ylast[(size_t)row * a.ldD + k]);
  }
}

hipError_t launch_head(const HeadArgs& a, int maxRows, hipStream_t s) {
  hipLaunchKernelGGL(head_kernel, dim3((maxRows + 3) / 4), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// adam_kernel: Adam::step + AdamOptimizer::apply_update (Network/Optimizer.cpp:61-108,122-160)
// with SMARTIES_NESTEROV_ADAM, SMARTIES_SAFE_ADAM, SMARTIES_ADAMW (Settings/Bund.h).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const DevScalars* sc = a.sc;
  const long long nStep = sc->nStep + 1;    // prepare_update incremented it before apply_update
  const float _eta = (float)((double)a.eta0 / (1 + (double)(float)nStep * a.epsAnneal));
  const float bt1 = (float)sc->adam_bt1, bt2 = (float)sc->adam_bt2;
  const float eta = _eta * sqrtf(1 - bt2) / (1 - bt1);
  const float B1 = 0.9f, B2 = 0.999f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    const float w = a.W[i];
    const float penal = -w * a.lambda;
    const float DW = a.fac * a.G[i];
    float m1 = B1 * a.M1[i] + (1 - B1) * DW;
    float m2 = B2 * a.M2[i] + (1 - B2) * DW * DW;
    const float numer = B1 * m1 + (1 - B1) * DW;
    m2 = m2 < m1 * m1 ? m1 * m1 : m2;
    const float ret = numer / (FLT_EPSILON + sqrtf(m2));
    a.M1[i] = m1; a.M2[i] = m2;
    a.W[i] = w + eta * (ret + penal);
  }
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t s) {
  const int blocks = (int)((a.n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// post_kernel (one workgroup): per-episode running aggregates in minibatch order
// (Episode::updateCumulative_atomic / updateValues_atomic, Episode.h:112-145), then
// MemoryProcessing::updateTrainingStatistics scalars (:187-259), updateCounters (:46-92),
// the Adam beta_t bookkeeping (Optimizer.cpp:155-160) and the step counter (Learner.cpp:130-133).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void aggValues(float* ag, float oldV, float oldADV, float V, float Q) {
  const float oldQ = oldADV + oldV;
  ag[AGG_SUMQ2] += Q * Q - oldQ * oldQ;
  ag[AGG_SUMQ] += Q - oldQ;
  ag[AGG_MAXQ] = fmaxf(ag[AGG_MAXQ], Q);
  ag[AGG_MINQ] = fminf(ag[AGG_MINQ], Q);
}

__global__ __launch_bounds__(256) void post_kernel(PostArgs a) {
  __shared__ long long sFarDelta;
  __shared__ unsigned sMaxAbs;
  DevScalars* sc = a.sc;
  const int tid = threadIdx.x, B = a.B;
  if (tid == 0) { sFarDelta = 0; sMaxAbs = 0u; }
  __syncthreads();
  if (a.mode & POST_AGG) {
    const float C = (float)sc->Cmax, invC = (float)sc->Cinv;
    for (int b = tid; b < B; b += 256) {
      const int e = a.bt.eid[b];
      if (b > 0 && a.bt.eid[b - 1] == e) continue;       // not the leader of this episode's run
      float* ag = a.rp.epAgg + (size_t)e * AGG_N;
      const float Nf = (float)a.rp.epN[e];
      const float invN = 1 / Nf;
      const long long before = farSteps(Nf, ag[AGG_FRACFAR]);
      for (int j = b; j < B && a.bt.eid[j] == e; ++j) {
        if (a.bt.nextOf[j] >= 0) {                         // setValues(t+1, Vnext) comes first
          const float Vn = a.bt.nextV[j];
          aggValues(ag, a.bt.oldNextV[j], a.bt.oldNextADV[j], Vn, Vn);
        }
        const float E = a.bt.newDQ[j], D = a.bt.newDKL[j], W = a.bt.newW[j];
        const float oW = a.bt.oldW[j], oE = a.bt.oldDQ[j];
        const float wasFar = (oW > C || oW < invC) ? 1.f : 0.f;
        const float isFar = (W > C || W < invC) ? 1.f : 0.f;
        ag[AGG_AVGKL] += invN * (D - a.bt.oldDKL[j]);
        ag[AGG_FRACFAR] += invN * (isFar - wasFar);
        ag[AGG_AVGSQERR] += invN * (E * E - oE * oE);
        ag[AGG_MAXABSERR] = fmaxf(ag[AGG_MAXABSERR], fabsf(E));
        const float Vf = a.bt.newV[j];
        aggValues(ag, a.bt.oldV[j], a.bt.oldADV[j], Vf, Vf);
      }
      const long long after = farSteps(Nf, ag[AGG_FRACFAR]);
      if (after != before) atomicAdd((unsigned long long*)&sFarDelta, (unsigned long long)(after - before));
      atomicMax(&sMaxAbs, __float_as_uint(fmaxf(ag[AGG_MAXABSERR], 0.f)));
    }
    __syncthreads();
    if (tid == 0) {
      sc->nFarTotal += sFarDelta;
      sc->maxAbsErrAll = fmaxf(sc->maxAbsErrAll, __uint_as_float(sMaxAbs));
      // updateTrainingStatistics: ReF-ER clip annealing for the NEXT sampling (:193-196)
      const long long k = sc->nGradSteps + 1;
      sc->Cmax = 1 + a.clipImpWeight / (1 + (double)k * a.epsAnneal);
      sc->Cinv = 1 / sc->Cmax;
      if (sc->Cmax <= 1) sc->nFarTotal = 0;
      sc->nFarStat = sc->nFarTotal; sc->cnt[2] = sc->nFarStat; sc->cnt[3] = sc->nTransitions;
    }
    __syncthreads();
  }
  if ((a.mode & (POST_BETA | POST_INIT)) && tid == 0) {
    // updateCounters (:46-92); with several replicas cnt[] holds the all-reduced counters
    const long long nFar = a.nRanks > 1 ? sc->cnt[2] : sc->nFarStat;
    const long long nStored = a.nRanks > 1 ? sc->cnt[3] : sc->nTransitions;
    const double fracOffPol = (double)nFar / (double)(nStored > 1 ? nStored : 1);
    const double nDataSize = fmax(a.maxObsGlobal, (double)nStored);
    const double learnRefer = 0.1 * a.batchGlobal / nDataSize;
    const double b0 = sc->beta, al0 = sc->alpha;
    const bool dec = fracOffPol > a.penalTol;
    sc->beta = dec ? (1 - fmin(learnRefer, b0)) * b0 : (1 - fmin(learnRefer, b0)) * b0 + fmin(learnRefer, 1 - b0);
    const bool decA = fabs(a.penalTol - fracOffPol) < 1e-3;
    sc->alpha = decA ? (1 - fmin(learnRefer, al0)) * al0 : (1 - fmin(learnRefer, al0)) * al0 + fmin(learnRefer, 1 - al0);
    if (a.mode & POST_BETA) {
      // stats.maxAbsError EMA (:239-240) uses the replica-local data size
      const double lrLoc = 0.1 * a.batchGlobal / fmax(a.maxObsGlobal, (double)sc->nTransitions);
      sc->maxAbsErrEMA += lrLoc * ((double)sc->maxAbsErrAll - sc->maxAbsErrEMA);
      sc->adam_bt1 *= 0.9; if (sc->adam_bt1 < (double)FLT_EPSILON) sc->adam_bt1 = 0;
      sc->adam_bt2 *= 0.999; if (sc->adam_bt2 < (double)FLT_EPSILON) sc->adam_bt2 = 0;
      sc->nStep += 1;
      sc->nGradSteps += 1;
    }
  }
}
hipError_t launch_post(const PostArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(post_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// episode_sweep_kernel: one thread per episode.
//   recompute=1: Episode::updateCumulative (Episode.cpp:213-242)
//   then computeRetrace backward scan (MemoryProcessing.cpp:23-44,391-400).
// Used on insert (count=1), by initializeLearner and every 1000th step (whole buffer).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void episode_sweep_kernel(EpisodeSweepArgs a) {
  __shared__ long long sFar[256];
  __shared__ float sMax[256];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  long long myFar = 0; float myMax = 0.f;
  if (idx < a.count) {
    const int e = a.eids ? a.eids[idx] : a.rp.posEid[idx];
    const long long off = a.rp.epOff[e];
    const int N = a.rp.epN[e];
    const bool term = a.rp.epTerm[e] != 0;
    const DevScalars* sc = a.sc;
    if (a.recompute) {
      const float C = (float)sc->Cmax, invC = (float)sc->Cinv;
      const int nd = N - 1;
      const float invN = 1 / (float)nd;
      long long nFarPol = 0;
      float sumE2 = 0, maxAE = -1e9f, maxQ = -1e9f, sumQ2 = 0, minQ = 1e9f, sumQ1 = 0, sumKL = 0;
      double totR = 0;
      for (int t = 0; t < nd; ++t) {
        const float w = a.rp.IMPW[off + t], dq = a.rp.DQ[off + t];
        if (w > C || w < invC) ++nFarPol;
        sumE2 += dq * dq; maxAE = fmaxf(maxAE, fabsf(dq));
        const float Q = a.rp.ADV[off + t] + a.rp.V[off + t];
        maxQ = fmaxf(maxQ, Q); minQ = fminf(minQ, Q); sumQ2 += Q * Q; sumQ1 += Q;
      }
      for (int t = 0; t < N; ++t) { totR += a.rp.R[off + t]; sumKL += a.rp.DKL[off + t]; }
      float* ag = a.rp.epAgg + (size_t)e * AGG_N;
      ag[AGG_FRACFAR] = invN * (float)nFarPol; ag[AGG_AVGSQERR] = invN * sumE2; ag[AGG_MAXABSERR] = maxAE;
      ag[AGG_SUMQ2] = sumQ2; ag[AGG_SUMQ] = sumQ1; ag[AGG_MAXQ] = maxQ; ag[AGG_MINQ] = minQ;
      ag[AGG_TOTR] = (float)totR; ag[AGG_AVGKL] = invN * sumKL;
      myFar = farSteps((float)N, ag[AGG_FRACFAR]);
      myMax = fmaxf(maxAE, 0.f);
    }
    const float gamma = a.gamma, lambda = a.lambda;
    const float rM = sc->rewMean, rS = sc->rewScale;
    float Q = term ? a.rp.RET[off + N - 1] : a.rp.V[off + N - 1];
    if (!term) a.rp.RET[off + N - 1] = Q;
    for (int t = N - 2; t >= 0; --t) {
      const float R = (float)((a.rp.R[off + t + 1] - (double)rM) * (double)rS);
      const float V = a.rp.V[off + t + 1], A = a.rp.ADV[off + t + 1];
      const float iw = a.rp.IMPW[off + t + 1];
      const float w = iw < 1.f ? iw : 1.f;
      Q = R + gamma * (V + lambda * w * (Q - A - V));
      a.rp.RET[off + t] = Q;
    }
  }
  if (a.recompute) {
    sFar[threadIdx.x] = myFar; sMax[threadIdx.x] = myMax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) { sFar[threadIdx.x] += sFar[threadIdx.x + s];
        sMax[threadIdx.x] = fmaxf(sMax[threadIdx.x], sMax[threadIdx.x + s]); }
      __syncthreads();
    }
    if (threadIdx.x == 0) { a.redNFar[blockIdx.x] = sFar[0]; a.redMaxAbs[blockIdx.x] = sMax[0]; }
  }
}
int sweep_blocks(int count) { return (count + 255) / 256; }
hipError_t launch_episode_sweep(const EpisodeSweepArgs& a, int nBlocks, hipStream_t s) {
  if (nBlocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(episode_sweep_kernel, dim3(nBlocks), dim3(256), 0, s, a);
  return hipGetLastError();
}
__global__ void sweep_finish_kernel(DevScalars* sc, const long long* redNFar, const float* redMaxAbs, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long f = 0; float m = 0.f;
  for (int i = 0; i < n; ++i) { f += redNFar[i]; m = fmaxf(m, redMaxAbs[i]); }
  sc->nFarTotal = sc->Cmax <= 1 ? 0 : f;
  sc->maxAbsErrAll = m;
  sc->nFarStat = sc->nFarTotal; sc->cnt[2] = sc->nFarStat; sc->cnt[3] = sc->nTransitions;
}
hipError_t launch_sweep_finish(DevScalars* sc, const long long* redNFar, const float* redMaxAbs, int nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(sweep_finish_kernel, dim3(1), dim3(64), 0, s, sc, redNFar, redMaxAbs, nBlocks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// moments: MemoryProcessing::updateRewardsStats (MemoryProcessing.cpp:94-185).
// One wavefront per episode (lanes stride over its transitions), fp64 accumulation
// (the reference uses long double on the host), deterministic two-stage reduction.
// ---------------------------------------------------------------------------
// Thread layout: column c = tid % (dS+1) (c < dS: state component, c == dS: reward), row lane
// r = tid / (dS+1); every thread owns one column, so sums are order-deterministic.
__global__ __launch_bounds__(256) void moments_partial_kernel(MomentsArgs a) {
  __shared__ double s1[256], s2[256];
  const int dS = a.dS, CW = dS + 1, RL = 256 / CW, tid = threadIdx.x;
  const int c = tid % CW, r = tid / CW;
  const bool active = r < RL;
  double sum = 0, sq = 0;
  const float rMean = a.sc->rewMean;
  const float sMean = (active && c < dS) ? a.rp.stMean[c] : 0.f;
  if (active) for (int p = blockIdx.x; p < a.nEpisodes; p += gridDim.x) {
    const int e = a.rp.posEid[p];
    const long long off = a.rp.epOff[e];
    const int nd = a.rp.epN[e] - 1;
    for (int j = r; j < nd; j += RL) {
      double d;
      if (c < dS) d = (double)(a.rp.S[(size_t)(off + j) * dS + c] - sMean);   // float - float
      else d = a.rp.R[off + j + 1] - (double)rMean;
      sum += d; sq += d * d;
    }
  }
  s1[tid] = sum; s2[tid] = sq;
  __syncthreads();
  if (tid < CW) {
    double t1 = 0, t2 = 0;
    for (int q = 0; q < RL; ++q) { t1 += s1[q * CW + tid]; t2 += s2[q * CW + tid]; }
    a.partial[(size_t)blockIdx.x * 2 * CW + tid] = t1;
    a.partial[(size_t)blockIdx.x * 2 * CW + CW + tid] = t2;
  }
}
// moments layout (MemoryProcessing.cpp:139-150): [sum s (dS) | sum s^2 (dS) | count | sum r | sum r^2]
__global__ __launch_bounds__(256) void moments_final_kernel(MomentsArgs a) {
  const int dS = a.dS, CW = dS + 1;
  for (int i = threadIdx.x; i < 2 * CW; i += 256) {
    double s = 0;
    for (int b = 0; b < a.nBlocks; ++b) s += a.partial[(size_t)b * 2 * CW + i];
    const int c = i % CW; const bool second = i >= CW;
    if (c < dS) a.moments[(second ? dS : 0) + c] = s;
    else a.moments[2 * dS + (second ? 2 : 1)] = s;
  }
  if (threadIdx.x == 0) a.moments[2 * dS] = (double)a.sc->nTransitions;
}
__global__ void moments_apply_kernel(MomentsArgs a) {
  DevScalars* sc = a.sc;
  const int dS = a.dS;
  const double learnR = a.learnrate / (1 + (double)sc->nGradSteps * a.epsAnneal);
  const double annealLearnR = fmin(1.0, a.rRateFac * learnR);
  const double Wt = a.bInit ? 1.0 : annealLearnR;
  if (!(Wt > 0)) return;
  const double count = a.moments[2 * dS];
  for (int i = threadIdx.x; i <= dS; i += blockDim.x) {
    const bool isRew = (i == dS);
    const double Evar = (isRew ? a.moments[2 * dS + 1] : a.moments[i]) / count;
    const double Evar2 = (isRew ? a.moments[2 * dS + 2] : a.moments[dS + i]) / count;
    float mean = isRew ? sc->rewMean : a.rp.stMean[i];
    float stdev = isRew ? sc->rewStd : a.rp.stStd[i];
    mean = (float)((double)mean + Wt * Evar);
    double variance = Evar2 - Evar * Evar * (2 * Wt - Wt * Wt);
    variance = fmax(variance, (double)FLT_EPSILON);
    stdev = (float)((double)stdev + Wt * (sqrt(variance) - (double)stdev));
    const float inv = 1 / stdev;
    if (isRew) { sc->rewMean = mean; sc->rewStd = stdev; sc->rewScale = inv; }
    else { a.rp.stMean[i] = mean; a.rp.stStd[i] = stdev; a.rp.stScale[i] = inv; }
  }
}
int moments_blocks(int nEpisodes) { int b = nEpisodes; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }
hipError_t launch_moments(const MomentsArgs& a, hipStream_t s) {
  if (a.dS + 1 > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moments_partial_kernel, dim3(a.nBlocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_moments_apply(const MomentsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(moments_apply_kernel, dim3(1), dim3(128), 0, s, a);
  return hipGetLastError();
}

__global__ void set_counts_kernel(DevScalars* sc, long long nT, long long nE, long long seenEps, long long seenSteps) {
  sc->nTransitions = nT; sc->nEpisodes = nE; sc->cnt[0] = seenEps; sc->cnt[1] = seenSteps;
  sc->cnt[3] = nT;   // cnt[2] keeps the far-policy count of the last statistics pass
}
// removal of an episode (MemoryBuffer::removeBackEpisode): its far-policy steps leave the total
__global__ void evict_kernel(DevScalars* sc, DevReplay rp, int eid) {
  const long long c = farSteps((float)rp.epN[eid], rp.epAgg[(size_t)eid * AGG_N + AGG_FRACFAR]);
  sc->nFarTotal -= c; if (sc->nFarTotal < 0) sc->nFarTotal = 0;
}
hipError_t launch_evict(DevScalars* sc, DevReplay rp, int eid, hipStream_t s) {
  hipLaunchKernelGGL(evict_kernel, dim3(1), dim3(1), 0, s, sc, rp, eid);
  return hipGetLastError();
}
hipError_t launch_set_counts(DevScalars* sc, long long nT, long long nE, long long seenEps, long long seenSteps, hipStream_t s) {
  hipLaunchKernelGGL(set_counts_kernel, dim3(1), dim3(1), 0, s, sc, nT, nE, seenEps, seenSteps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stats_kernel (one workgroup): the reduction over all episodes of
// MemoryProcessing::updateTrainingStatistics (:209-258), on demand (logging surface).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_kernel(DevScalars* sc, DevReplay rp, int nEp, double* out) {
  __shared__ double sd[5][256];
  __shared__ float sf[2][256];
  const int tid = threadIdx.x;
  double sumDKL = 0, sumE2 = 0, sumQ2 = 0, sumQ1 = 0, sumR = 0;
  float maxQ = -1e9f, minQ = 1e9f;
  for (int p = tid; p < nEp; p += 256) {
    const int e = rp.posEid[p];
    const float* ag = rp.epAgg + (size_t)e * AGG_N;
    const float Nf = (float)rp.epN[e];
    sumDKL += (double)(Nf * ag[AGG_AVGKL]); sumE2 += (double)(Nf * ag[AGG_AVGSQERR]);
    sumQ2 += (double)ag[AGG_SUMQ2]; sumQ1 += (double)ag[AGG_SUMQ]; sumR += (double)ag[AGG_TOTR];
    maxQ = fmaxf(maxQ, ag[AGG_MAXQ]); minQ = fminf(minQ, ag[AGG_MINQ]);
  }
  sd[0][tid] = sumDKL; sd[1][tid] = sumE2; sd[2][tid] = sumQ2; sd[3][tid] = sumQ1; sd[4][tid] = sumR;
  sf[0][tid] = maxQ; sf[1][tid] = minQ;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      for (int q = 0; q < 5; ++q) sd[q][tid] += sd[q][tid + s];
      sf[0][tid] = fmaxf(sf[0][tid], sf[0][tid + s]); sf[1][tid] = fminf(sf[1][tid], sf[1][tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double nData = (double)sc->nTransitions;
    out[0] = sd[0][0] / nData;                 // avgKLdivergence
    out[1] = sd[1][0] / nData;                 // avgSquaredErr
    out[2] = sc->maxAbsErrEMA;                 // maxAbsError
    out[3] = sd[4][0] / (double)nEp;           // avgReturn
    const double avgQ = sd[3][0] / nData;
    out[4] = avgQ;
    out[5] = sqrt(fmax(sd[2][0] / nData - avgQ * avgQ, 1e-16));   // stdevQ
    out[6] = (double)sf[1][0]; out[7] = (double)sf[0][0];          // minQ, maxQ
    out[8] = (double)sc->nFarStat;
  }
}
hipError_t launch_stats(DevScalars* sc, DevReplay rp, int nEpisodes, double* out, hipStream_t s) {
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(256), 0, s, sc, rp, nEpisodes, out);
  return hipGetLastError();
}

}  // namespace hl
