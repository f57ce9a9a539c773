// smarties_amd/csrc/conv.hip -- convolutional preprocessing layers (BASELINE config 5: RACER_atari.json) as implicit
// GEMMs on fp32 MFMA, and the minibatch gather for states with appended past observations.
//
// Reference: Conv2DLayer<SoftSign, ...> (Network/Layers/Layer_Conv2D.h:29-232): image [InC][InY][InX], filter
// K[KnC][InC][KnY][KnX], ONE BIAS PER OUTPUT ELEMENT ([KnC][OpY][OpX]), no padding in the shipped shapes
// (Network/Builder.cpp:189-203); Episode::standardizedState (ReplayMemory/Episode.h:172-183) for the stacked input.
// Parameters and activations keep the reference's layouts (so checkpoints, the dense layer behind the last convolution
// and the parametric residual over its first outputs need no permutation); nothing is materialised as an im2col matrix:
//
//   forward   Y[c][(b,p)]   = sum_k  K[c][k] * in[(b,p) -> patch element k]        k = (ic, fy, fx)
//             MFMA A = filter rows (LDS, padded pitch), B = patch elements gathered from the input image through a
//             per-k offset table; a wavefront owns 32 output positions x all channels; the 16-lane rows of the
//             result tile are consecutive output positions, so X / Y stores are 64-byte runs per channel
//   dX        dIn[ic][(b,q)] = sum_{c,fy,fx} K[c][ic][fy][fx] * D[(b, c, (q - f) / S)]   (valid positions only), then
//             x act'(X_in): the gather form of Layer_Conv2D.h:117-138 -- no atomics, fixed summation order
//   dW        dK[c][k]      = sum_{(b,p)} D[b][c][p] * in[(b,p) -> k]: the reduction over batch x positions is cut into
//             chunks (one workgroup each, four waves interleaved), partial tiles go to a scratch array and are summed
//             in chunk order by conv_reduce_adam_kernel, which also applies Adam -- bit-deterministic
//   bias      column sums of D over the batch: RED_COL problems of the common weight-gradient launch (gemm16.hip)
#include "dev_common.h"
#include "dw_wide_dev.h"

namespace hl {

__device__ __forceinline__ void convLdsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // workgroup barrier that settles LDS traffic only: global loads stay in flight (what __syncthreads compiles to for gfx950 as well; spelled out where the code relies on it)
__device__ __forceinline__ float softsignEval(float x) { return x / (1 + fabsf(x)); }
__device__ __forceinline__ float softsignDiff(float x) { const float d = 1 + fabsf(x); return 1 / (d * d); }

// ---------------------------------------------------------------------------------------------------------------
// gather: X0[row][j dS + i] = (S[slot(row) - min(j, t(row))][i] - mean[i]) * scale[i],  j = 0 .. nAppendedObs
// rows < B: the sampled steps; rows >= B: the truncated next states s_{t+1} (MemoryBuffer.cpp:413-429)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stack_gather_kernel(StackGatherArgs a) {
  const int row = blockIdx.y;
  const DevScalars* sc = a.sc;
  if (row >= sc->nRows[a.parity]) return;
  const int b = row < a.B ? row : a.bt.nextSrc[row - a.B];
  const long long slot = a.bt.slot[b] + (row < a.B ? 0 : 1);
  const int t = a.bt.t[b] + (row < a.B ? 0 : 1);
  const int dS = a.dS, dIn = dS * (1 + a.nApp);
  if ((dS & 3) == 0) {                                        // 16-byte path (frames of 84 x 84: 7056 floats)
    const int q4 = dS >> 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < (dIn >> 2); idx += gridDim.x * 256) {
      const int j = idx / q4, i = idx - j * q4;
      const int back = j < t ? j : t;                         // steps before the first repeat the first
      const f32x4 v = reinterpret_cast<const f32x4*>(a.rp.S + (size_t)(slot - back) * dS)[i];
      const f32x4 m = reinterpret_cast<const f32x4*>(a.rp.stMean)[i], sc4 = reinterpret_cast<const f32x4*>(a.rp.stScale)[i];
      f32x4 o; o[0] = (v[0] - m[0]) * sc4[0]; o[1] = (v[1] - m[1]) * sc4[1]; o[2] = (v[2] - m[2]) * sc4[2]; o[3] = (v[3] - m[3]) * sc4[3];
      reinterpret_cast<f32x4*>(a.X0 + (size_t)row * a.ldX0)[idx] = o;
    }
    return;
  }
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < dIn; idx += gridDim.x * 256) {
    const int j = idx / dS, i = idx - j * dS;
    const int back = j < t ? j : t;
    a.X0[(size_t)row * a.ldX0 + idx] = (a.rp.S[(size_t)(slot - back) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
  }
}
// short rows (a few dozen state components): 256 / P rows per workgroup, P = the power of two above the row length -- with a workgroup
// per row a large batch is 32768 workgroups of 17 busy threads (15.5 us at 16384 samples)
__global__ __launch_bounds__(256) void stack_gather_rows_kernel(StackGatherArgs a, int P) {
  const int row = blockIdx.x * (256 / P) + (int)threadIdx.x / P, idx = (int)threadIdx.x & (P - 1);
  const DevScalars* sc = a.sc;
  const int dS = a.dS, dIn = dS * (1 + a.nApp);
  if (row >= sc->nRows[a.parity] || idx >= dIn) return;
  const int b = row < a.B ? row : a.bt.nextSrc[row - a.B];
  const long long slot = a.bt.slot[b] + (row < a.B ? 0 : 1);
  const int t = a.bt.t[b] + (row < a.B ? 0 : 1);
  const int j = idx / dS, i = idx - j * dS;
  const int back = j < t ? j : t;
  a.X0[(size_t)row * a.ldX0 + idx] = (a.rp.S[(size_t)(slot - back) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
}
hipError_t launch_stack_gather(const StackGatherArgs& a, int maxRows, hipStream_t s) {
  const int dIn = a.dS * (1 + a.nApp);
  if (dIn <= 64 && maxRows >= 1024) {
    int P = 4; while (P < dIn) P <<= 1;
    hipLaunchKernelGGL(stack_gather_rows_kernel, dim3((maxRows + 256 / P - 1) / (256 / P)), dim3(256), 0, s, a, P);
    return hipGetLastError();
  }
  int bx = (dIn / 4 + 255) / 256; if (bx > 32) bx = 32; if (bx < 1) bx = 1;
  hipLaunchKernelGGL(stack_gather_kernel, dim3(bx, maxRows), dim3(256), 0, s, a);
  return hipGetLastError();
}

// State variables beside the image (Approximator.cpp:249-259: a second input layer behind the conv stack, glued to its output by a
// JoinLayer -- the later layer first, Layers.h:289-299): columns [col0, col0 + n) of the stacked rows go in front of the last
// convolution's outputs, dst[row][0 .. n)
__global__ __launch_bounds__(256) void extras_copy_kernel(const DevScalars* sc, int parity, const float* X0, int ldX0, int col0, int n, float* dst, int ldDst) {
  const int row = blockIdx.y;
  if (row >= sc->nRows[parity]) return;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) dst[(size_t)row * ldDst + e] = X0[(size_t)row * ldX0 + col0 + e];
}
hipError_t launch_extras_copy(const DevScalars* sc, int parity, const float* X0, int ldX0, int col0, int n, float* dst, int ldDst, int maxRows, hipStream_t s) {
  hipLaunchKernelGGL(extras_copy_kernel, dim3(std::min((n + 255) / 256, 8), maxRows), dim3(256), 0, s, sc, parity, X0, ldX0, col0, n, dst, ldDst);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  A workgroup = 4 wavefronts = (4 / CT) tiles of 16 output positions x CT tiles of 16 channels, one
// (position tile, channel tile) per wavefront.  These layers are small (0.2 - 0.8 MFLOP per sample), so the kernels are
// built for latency: NK > 0 = the number of MFMA steps is known at compile time and EVERY gathered patch element of the
// wavefront is requested before the first MFMA (one exposed memory round trip instead of one per few steps); NK = 0 is
// the generic loop for other shapes.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int convPad4(int k) { return (k + 3) & ~3; }
__host__ __device__ inline size_t convFwdLds(const ConvGeo& g, int CT) {
  const int Kp = convPad4(g.K);
  return (size_t)CT * 16 * (Kp + 4) * 4 + (size_t)Kp * 4;
}

// Filters in the layouts the kernels stage into LDS, written once per step (the weights change only in the Adam pass):
//   Wf[l]: [CT 16][Kp + 4]   filter rows, zero padded (forward: A operand rows = output channels)
//   Wx[l]: [IT 16][KKp + 4]  Wx[ic][(c, fy, fx)] = K[c][ic][fy][fx], zero padded (dX: A operand rows = input channels)
// so that a workgroup's staging is a flat 16-byte copy with every load in flight at once.
__host__ __device__ inline int convWfFloats(const ConvGeo& g) { return ((g.KnC + 15) & ~15) * (convPad4(g.K) + 4); }
// strided layers whose filter and input sizes are multiples of the stride: dX runs per PARITY CLASS of the input position
// (iy mod S, ix mod S) -- only the filter taps fy = iy (mod S), fx = ix (mod S) reach such a position, 1 / S^2 of them
__host__ __device__ inline bool convStrided(const ConvGeo& g) { return g.S > 1 && g.KnY % g.S == 0 && g.KnX % g.S == 0 && g.InY % g.S == 0 && g.InX % g.S == 0; }
__host__ __device__ inline int convClassK(const ConvGeo& g) { return g.KnC * (g.KnY / g.S) * (g.KnX / g.S); }
__host__ __device__ inline int convWxFloats(const ConvGeo& g) {
  const int rows = (g.InC + 15) & ~15;
  if (convStrided(g)) return g.S * g.S * rows * (convPad4(convClassK(g)) + 4);      // Wx[class][ic][(c, ty, tx)]
  return rows * (convPad4(g.KnC * g.KnY * g.KnX) + 4);
}
__global__ __launch_bounds__(256) void conv_prep_kernel(ConvArgs a) {
  int l = 0, i = blockIdx.x * 256 + threadIdx.x;
  for (; l < a.nL; ++l) { const int n = convWfFloats(a.L[l]) + (l > 0 ? convWxFloats(a.L[l]) : 0); if (i < n) break; i -= n; }
  if (l >= a.nL) return;
  const ConvGeo& g = a.L[l];
  const float* Wl = a.W + g.indW;
  const int nf = convWfFloats(g);
  if (i < nf) {
    const int ldK = convPad4(g.K) + 4, c = i / ldK, k = i - c * ldK;
    g.Wf[i] = (c < g.KnC && k < g.K) ? Wl[(size_t)c * g.K + k] : 0.f;
  } else {
    i -= nf;
    if (convStrided(g)) {
      const int S = g.S, TY = g.KnY / S, TX = g.KnX / S, KKc = g.KnC * TY * TX, ld = convPad4(KKc) + 4, rows = (g.InC + 15) & ~15;
      const int cls = i / (rows * ld), rem = i - cls * rows * ld, ic = rem / ld, kc = rem - ic * ld;
      float w = 0.f;
      if (ic < g.InC && kc < KKc) {
        const int c = kc / (TY * TX), t = kc - c * TY * TX, ty = t / TX, tx = t - ty * TX;
        const int fy = cls / S + S * ty, fx = cls % S + S * tx;
        w = Wl[(((size_t)c * g.InC + ic) * g.KnY + fy) * g.KnX + fx];
      }
      g.Wx[i] = w;
      return;
    }
    const int fsz = g.KnY * g.KnX, KK = g.KnC * fsz, ldKK = convPad4(KK) + 4, ic = i / ldKK, kk = i - ic * ldKK;
    float w = 0.f;
    if (ic < g.InC && kk < KK) { const int c = kk / fsz, f = kk - c * fsz; w = Wl[((size_t)c * g.InC + ic) * fsz + f]; }
    g.Wx[i] = w;
  }
}
hipError_t launch_conv_prep(const ConvArgs& a, hipStream_t s) {
  long long n = 0;
  for (int l = 0; l < a.nL; ++l) n += convWfFloats(a.L[l]) + (l > 0 ? convWxFloats(a.L[l]) : 0);
  hipLaunchKernelGGL(conv_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}
long long conv_prep_floats(const ConvGeo& g, int which) { return which == 0 ? convWfFloats(g) : convWxFloats(g); }

// flat 16-byte copy global -> LDS, every load of the thread issued before its first store
template <int MAXQ> __device__ __forceinline__ void stageFlat(float* dst, const float* src, int nFloats) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src); f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  const int n4 = nFloats >> 2;
  for (int q0 = 0; q0 < n4; q0 += 256 * MAXQ) {
    f32x4 v[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) { const int i = q0 + threadIdx.x + 256 * q; v[q] = i < n4 ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) { const int i = q0 + threadIdx.x + 256 * q; if (i < n4) d4[i] = v[q]; }
  }
}

// KS = 4 (small layers: few position tiles): the four wavefronts share ONE tile's reduction, the channel tile comes from
// blockIdx.y -- four times the workgroups, a quarter of the chain per wavefront; partial tiles meet in LDS in wave order
template <int CT, int NK, int KS = 1>     // channel tiles per workgroup (1, 2, 4); MFMA steps (0: run-time); waves per tile
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a, int l) {
  static_assert(KS == 1 || (CT == 1 && NK > 0 && NK % KS == 0), "split tiles: one channel tile per workgroup");
  __shared__ float sRed[KS > 1 ? 4 : 1][256];
  const int ctBase = KS > 1 ? blockIdx.y : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo g = a.L[l];
  const int K = g.K, Kp = convPad4(K), ldK = Kp + 4, P = g.P;
  float* Ws = reinterpret_cast<float*>(smem);                         // [CT*16][ldK]
  int* kOff = reinterpret_cast<int*>(Ws + (size_t)CT * 16 * ldK);     // [Kp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  constexpr int PW = 4 / (CT * KS);                                    // position tiles per workgroup
  const int nRows = a.sc->nRows[a.parity];                            // (tested behind the filter requests below: they do not wait for it)
  // NK > 0: the filter rows are only REQUESTED here and stored to LDS behind the patch gathers below (the offset table needs no global
  // data): the workgroup's two round trips -- filters through the L2, the image rows from the launch in front -- overlap
  constexpr int WQ = NK > 0 ? (CT * 16 * (NK + 1) + 255) / 256 : 1;
  f32x4 wv[WQ];
  if constexpr (NK > 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(g.Wf + (size_t)ctBase * 16 * ldK); const int n4 = (CT * 16 * ldK) >> 2;
#pragma unroll
    for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; wv[q] = i < n4 ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  const unsigned R = (unsigned)nRows * (unsigned)P;                   // (rows x positions < 2^31, checked at creation)
  if (blockIdx.x * PW * 16u >= R) return;                             // whole workgroup beyond the minibatch
  if constexpr (NK == 0) stageFlat<5>(Ws, g.Wf + (size_t)ctBase * 16 * ldK, CT * 16 * ldK);
  const int ct = wave % CT, ks = (wave / CT) % KS;
  const unsigned tile = blockIdx.x * PW + wave / (CT * KS);
  // this lane's output position and the origin of its patch in the input image (the index arithmetic overlaps the
  // weight fetch)
  const unsigned r = tile * 16 + li;
  const bool ok = r < R;
  const unsigned rr = ok ? r : 0;
  const int bb = (int)(rr / (unsigned)P), pp = (int)(rr - (unsigned)bb * (unsigned)P);
  const int oy = pp / g.OpX, ox = pp - oy * g.OpX;
  const float* inRow = g.in + (long long)bb * g.ldIn + (long long)oy * g.S * g.InX + ox * g.S;
  // the biases of this lane's four outputs, requested now: behind the reduction's barrier they cost a round trip of their own
  const float* Bl = a.W + g.indB;
  float bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int ch = (ctBase + wave % CT) * 16 + lc * 4 + q; bq[q] = (ok && ch < g.KnC) ? Bl[(size_t)ch * P + pp] : 0.f; }
  for (int k = tid; k < Kp; k += 256) {
    int off = 0;
    if (k < K) { const int ic = k / (g.KnY * g.KnX), f = k - ic * g.KnY * g.KnX, fy = f / g.KnX, fx = f - fy * g.KnX;
      off = ic * g.InY * g.InX + fy * g.InX + fx; }
    kOff[k] = off;
  }
  if constexpr (NK > 0) convLdsBarrier(); else __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* wRow = Ws + (ct * 16 + li) * ldK + lc;
  if constexpr (NK > 0) {
    constexpr int NS = NK / KS;
    float bv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bv[s] = inRow[kOff[4 * (ks * NS + s) + lc]];
    __builtin_amdgcn_sched_barrier(0);          // all gathers are in flight before the first MFMA
    {
      f32x4* d4 = reinterpret_cast<f32x4*>(Ws); const int n4 = (CT * 16 * ldK) >> 2;
#pragma unroll
      for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; if (i < n4) d4[i] = wv[q]; }
    }
    convLdsBarrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float av = wRow[4 * (ks * NS + s)];
      if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc0, 0, 0, 0);
    }
    if constexpr (KS > 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sRed[wave][q * 64 + lane] = acc0[q] + acc1[q];
      __syncthreads();
      if (ks != 0) return;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = sRed[0][q * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) v += sRed[w][q * 64 + lane];
        acc0[q] = v; acc1[q] = 0.f;
      }
    }
  } else {
    constexpr int UN = 8;
    for (int s0 = 0; s0 < Kp / 4; s0 += UN) {
      float av[UN], bv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int s = s0 + u; const bool kin = 4 * s < Kp;
        av[u] = kin ? wRow[4 * s] : 0.f;
        bv[u] = kin ? inRow[kOff[4 * s + lc]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
      }
    }
  }
  if (!ok) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ch = (ctBase + ct) * 16 + lc * 4 + q;
    if (ch < g.KnC) {
      const float x = (acc0[q] + acc1[q]) + bq[q];
      const size_t o = (size_t)bb * g.ldOut + (size_t)ch * P + pp;
      g.X[o] = x; g.Y[o] = softsignEval(x);
    }
  }
}

template <int CT, int NK> static hipError_t launchConvFwdT(const ConvArgs& a, int l, long long R, hipStream_t s) {
  if constexpr (NK > 0) {
    if (R <= 16 * 1024) {          // few position tiles: split every tile's reduction over the four wavefronts
      const size_t lds1 = convFwdLds(a.L[l], 1);
      hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_fwd_kernel<1, NK, 4>), lds1);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((conv_fwd_kernel<1, NK, 4>), dim3((unsigned)((R + 15) / 16), (a.L[l].KnC + 15) / 16), dim3(256), lds1, s, a, l);
      return hipGetLastError();
    }
  }
  const size_t lds = convFwdLds(a.L[l], CT);
  constexpr int PW = 4 / CT;
  const int blocks = (int)((R + 16 * PW - 1) / (16 * PW));
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_fwd_kernel<CT, NK>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_fwd_kernel<CT, NK>), dim3(blocks), dim3(256), lds, s, a, l);
  return hipGetLastError();
}
template <int CT> static hipError_t launchConvFwdC(const ConvArgs& a, int l, long long R, hipStream_t s) {
  const int nk = convPad4(a.L[l].K) / 4;
  if (nk == 64) return launchConvFwdT<CT, 64>(a, l, R, s);       // 4 x 8 x 8, 16 x 4 x 4 patches
  if (nk == 72) return launchConvFwdT<CT, 72>(a, l, R, s);       // 8 x 6 x 6, 32 x 3 x 3
  return launchConvFwdT<CT, 0>(a, l, R, s);
}
hipError_t launch_conv_forward(const ConvArgs& a, int l, int maxRows, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const long long R = (long long)maxRows * g.P;
  const int CT = (g.KnC + 15) / 16;
  if (CT == 1) return launchConvFwdC<1>(a, l, R, s);
  if (CT == 2) return launchConvFwdC<2>(a, l, R, s);
  if (CT <= 4) return launchConvFwdC<4>(a, l, R, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// dX: gradient w.r.t. the input image of layer l, times act' of the layer below -> D of layer l - 1.  Same workgroup
// shape as the forward kernel: one (16 input positions, 16 input channels) tile per wavefront, every gathered delta
// requested before the first MFMA when the number of steps is a compile-time constant.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t convDxLds(const ConvGeo& g, int IT) {
  const int KKp = convPad4(g.KnC * g.KnY * g.KnX);
  return (size_t)IT * 16 * (KKp + 4) * 4 + (size_t)KKp * 4;
}
// KS waves share the reduction of one tile (NK > 0): the chain per wavefront is NK / KS steps, partial tiles meet in LDS in wave
// order (fixed summation order), the first wave of a tile applies act' and stores
template <int IT, int NK>     // input-channel tiles per workgroup (NK > 0: 1, the tile of blockIdx.y); MFMA steps (0: run-time)
__global__ __launch_bounds__(256) void conv_dx_kernel(ConvArgs a, int l, DenseRide ride) {
  constexpr int KS = NK > 0 ? 4 / IT : 1;      // waves per tile
  const int itBase = NK > 0 ? blockIdx.y : 0;
  __shared__ float sRed[4][256];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // behind the launch's own workgroups: weight-gradient tiles of the dense layers (their deltas were final before this launch; a
  // launch of this kind leaves most of the chip idle) -- dw_wide_dev.h, one workgroup per tile
  if (ride.tile1 > ride.tile0 && (int)blockIdx.x >= ride.own) {
    const int gt = ride.tile0 + ((int)blockIdx.x - ride.own) * (int)gridDim.y + (int)blockIdx.y;
    if (gt < ride.tile1) dwWideBody<16>(ride.probs, ride.nProbs, ride.tile1, 1, nullptr, nullptr, a.sc, ride.hyp, gt, smem);
    return;
  }
  const ConvGeo g = a.L[l];
  const ConvGeo gp = a.L[l - 1];                    // the layer whose outputs are this layer's inputs
  const int KK = g.KnC * g.KnY * g.KnX, KKp = convPad4(KK), ldKK = KKp + 4, P = g.P, Pin = g.InY * g.InX;
  float* Wx = reinterpret_cast<float*>(smem);                          // [IT*16][ldKK]   Wx[ic][(c, fy, fx)]
  int* kTab = reinterpret_cast<int*>(Wx + (size_t)IT * 16 * ldKK);     // [KKp]   c * P | fy << 20 | fx << 26   (-1: padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  constexpr int PW = 4 / (IT * KS);
  const unsigned R = (unsigned)a.B * (unsigned)Pin;
  // NK > 0: the filter rows are only REQUESTED here and stored to LDS behind the delta gathers below (the offset table needs no
  // global data), so the two round trips of a workgroup -- filters through the L2, deltas from the launch in front -- overlap
  constexpr int WQ = NK > 0 ? (IT * 16 * (NK + 1) + 255) / 256 : 1;
  f32x4 wv[WQ];
  if constexpr (NK > 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(g.Wx + (size_t)itBase * 16 * ldKK); const int n4 = (IT * 16 * ldKK) >> 2;
#pragma unroll
    for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; wv[q] = i < n4 ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  } else stageFlat<5>(Wx, g.Wx + (size_t)itBase * 16 * ldKK, IT * 16 * ldKK);
  const int it = wave % IT, ks = (wave / IT) % KS;
  const unsigned tile = blockIdx.x * PW + wave / (IT * KS);
  const unsigned r = tile * 16 + li;
  const bool ok = r < R;
  const unsigned rr = ok ? r : 0;
  const int bb = (int)(rr / (unsigned)Pin), qq = (int)(rr - (unsigned)bb * (unsigned)Pin);
  const int iy = qq / g.InX, ix = qq - iy * g.InX;
  const float* dRow = g.D + (long long)bb * g.ldOut;
  const int fsz = g.KnY * g.KnX;
  // pre-activations of this lane's four outputs (for act'), requested now rather than behind the reduction
  float xq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int ic = (itBase + wave % IT) * 16 + lc * 4 + q; xq[q] = (ok && ic < g.InC) ? gp.X[(size_t)bb * gp.ldOut + (size_t)ic * Pin + qq] : 0.f; }
  for (int kk = tid; kk < KKp; kk += 256) {
    int v = -1;
    if (kk < KK) { const int c = kk / fsz, f = kk - c * fsz, fy = f / g.KnX, fx = f - fy * g.KnX; v = (c * P) | (fy << 20) | (fx << 26); }
    kTab[kk] = v;
  }
  if constexpr (NK > 0) convLdsBarrier(); else __syncthreads();      // (NK > 0: the table only; the filter loads stay in flight)
  const int S = g.S, sh = S == 1 ? 0 : (S == 2 ? 1 : (S == 4 ? 2 : 3));
  auto gatherD = [&](int kk) -> float {       // D[(b, c, (q - f) / S)] where that output position exists, else 0
    const int tab = kTab[kk];
    const int offC = tab & 0xFFFFF, fy = (tab >> 20) & 63, fx = (tab >> 26) & 31;
    const int oyS = iy - fy, oxS = ix - fx;
    const int oy = oyS >> sh, ox = oxS >> sh;
    const bool v = tab >= 0 && ok && oyS >= 0 && oxS >= 0 && ((oyS | oxS) & (S - 1)) == 0 && oy < g.OpY && ox < g.OpX;
    return v ? dRow[offC + oy * g.OpX + ox] : 0.f;
  };
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* wRow = Wx + (it * 16 + li) * ldKK + lc;
  if constexpr (NK > 0) {
    constexpr int NS = NK / KS;
    static_assert(NK % KS == 0, "steps divide over the waves");
    float bv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bv[s] = gatherD(4 * (ks * NS + s) + lc);
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4* d4 = reinterpret_cast<f32x4*>(Wx); const int n4 = (IT * 16 * ldKK) >> 2;
#pragma unroll
      for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; if (i < n4) d4[i] = wv[q]; }
    }
    convLdsBarrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float av = wRow[4 * (ks * NS + s)];
      if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc0, 0, 0, 0);
    }
    if constexpr (KS > 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sRed[wave][q * 64 + lane] = acc0[q] + acc1[q];
      __syncthreads();
      if (ks != 0) return;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = sRed[wave][q * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) v += sRed[wave + w * IT][q * 64 + lane];
        acc0[q] = v; acc1[q] = 0.f;
      }
    }
  } else {
    constexpr int UN = 8;
    for (int s0 = 0; s0 < KKp / 4; s0 += UN) {
      float av[UN], bv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int s = s0 + u; const bool kin = 4 * s < KKp;
        av[u] = kin ? wRow[4 * s] : 0.f;
        bv[u] = kin ? gatherD(4 * s + lc) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
      }
    }
  }
  if (!ok) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ic = (itBase + it) * 16 + lc * 4 + q;
    if (ic < g.InC) {
      const size_t o = (size_t)bb * gp.ldOut + (size_t)ic * Pin + qq;
      gp.D[o] = (acc0[q] + acc1[q]) * softsignDiff(xq[q]);
    }
  }
}
// the same for a strided layer, one parity class of input positions per workgroup: tiles are laid out class-major
// (class, sample, position inside the class), NK = KnC (KnY / S)(KnX / S) / 4 steps instead of KnC KnY KnX / 4
template <int IT, int NK, int KSP = 0>      // KSP: wavefronts per tile (0: four for the instantiated shapes; 1: a tile per wavefront -- launches with thousands of tiles)
__global__ __launch_bounds__(256) void conv_dxs_kernel(ConvArgs a, int l, unsigned tilesPerClass) {
  constexpr int KS = KSP > 0 ? KSP : (NK > 0 ? 4 / IT : 1);
  __shared__ float sRed[4][256];
  const int itBase = NK > 0 ? blockIdx.y : 0;
  const int rowsAll = (a.L[l].InC + 15) & ~15;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo g = a.L[l];
  const ConvGeo gp = a.L[l - 1];
  const int S = g.S, TY = g.KnY / S, TX = g.KnX / S, KKc = g.KnC * TY * TX, KKp = convPad4(KKc), ld = KKp + 4, P = g.P;
  const int CX = g.InX / S, Pc = (g.InY / S) * CX, Pin = g.InY * g.InX;
  float* Wx = reinterpret_cast<float*>(smem);                          // [IT*16][ld]   this class's Wx[ic][(c, ty, tx)]
  int* kTab = reinterpret_cast<int*>(Wx + (size_t)IT * 16 * ld);       // [KKp]   c * P | ty << 20 | tx << 26   (-1: padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  constexpr int PW = 4 / (IT * KS);
  const unsigned Rc = (unsigned)a.B * (unsigned)Pc;
  const unsigned tile = blockIdx.x * PW + wave / (IT * KS);
  const int cls = (int)(tile / tilesPerClass);                         // uniform over the workgroup (tilesPerClass is a multiple of PW)
  const unsigned tIn = tile - (unsigned)cls * tilesPerClass;
  constexpr int WQ = NK > 0 ? (IT * 16 * (NK + 1) + 255) / 256 : 1;      // (as conv_dx_kernel: requested here, stored behind the delta gathers)
  f32x4 wv[WQ];
  if constexpr (NK > 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(g.Wx + ((size_t)cls * rowsAll + (size_t)itBase * 16) * ld); const int n4 = (IT * 16 * ld) >> 2;
#pragma unroll
    for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; wv[q] = i < n4 ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  } else stageFlat<5>(Wx, g.Wx + ((size_t)cls * rowsAll + (size_t)itBase * 16) * ld, IT * 16 * ld);
  const int it = wave % IT, ks = (wave / IT) % KS;
  const unsigned r = tIn * 16 + li;
  const bool ok = r < Rc;
  const unsigned rr = ok ? r : 0;
  const int bb = (int)(rr / (unsigned)Pc), j = (int)(rr - (unsigned)bb * (unsigned)Pc);
  const int jy = j / CX, jx = j - jy * CX;
  const int qq = (cls / S + S * jy) * g.InX + (cls % S + S * jx);
  const float* dRow = g.D + (long long)bb * g.ldOut;
  for (int kc = tid; kc < KKp; kc += 256) {
    int v = -1;
    if (kc < KKc) { const int c = kc / (TY * TX), t = kc - c * TY * TX, ty = t / TX, tx = t - ty * TX; v = (c * P) | (ty << 20) | (tx << 26); }
    kTab[kc] = v;
  }
  if constexpr (NK > 0) convLdsBarrier(); else __syncthreads();
  auto gatherD = [&](int kc) -> float {       // D[(b, c, (jy - ty, jx - tx))] where that output position exists, else 0
    const int tab = kTab[kc];
    const int offC = tab & 0xFFFFF, oy = jy - ((tab >> 20) & 63), ox = jx - ((tab >> 26) & 31);
    const bool v = tab >= 0 && ok && (unsigned)oy < (unsigned)g.OpY && (unsigned)ox < (unsigned)g.OpX;
    return v ? dRow[offC + oy * g.OpX + ox] : 0.f;
  };
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* wRow = Wx + (it * 16 + li) * ld + lc;
  if constexpr (NK > 0) {
    constexpr int NS = NK / KS;
    static_assert(NK % KS == 0, "steps divide over the waves");
    float bv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bv[s] = gatherD(4 * (ks * NS + s) + lc);
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4* d4 = reinterpret_cast<f32x4*>(Wx); const int n4 = (IT * 16 * ld) >> 2;
#pragma unroll
      for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; if (i < n4) d4[i] = wv[q]; }
    }
    convLdsBarrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float av = wRow[4 * (ks * NS + s)];
      if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[s], acc0, 0, 0, 0);
    }
    if constexpr (KS > 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sRed[wave][q * 64 + lane] = acc0[q] + acc1[q];
      __syncthreads();
      if (ks != 0) return;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = sRed[wave][q * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) v += sRed[wave + w * IT][q * 64 + lane];
        acc0[q] = v; acc1[q] = 0.f;
      }
    }
  } else {
    constexpr int UN = 8;
    for (int s0 = 0; s0 < KKp / 4; s0 += UN) {
      float av[UN], bv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int s = s0 + u; const bool kin = 4 * s < KKp;
        av[u] = kin ? wRow[4 * s] : 0.f;
        bv[u] = kin ? gatherD(4 * s + lc) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
      }
    }
  }
  if (!ok) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ic = (itBase + it) * 16 + lc * 4 + q;
    if (ic < g.InC) {
      const size_t o = (size_t)bb * gp.ldOut + (size_t)ic * Pin + qq;
      gp.D[o] = (acc0[q] + acc1[q]) * softsignDiff(gp.X[o]);
    }
  }
}
template <int IT, int NK, int KSP = 0> static hipError_t launchConvDxsT(const ConvArgs& a, int l, int ity, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const int KKp = convPad4(convClassK(g));
  const size_t lds = (size_t)IT * 16 * (KKp + 4) * 4 + (size_t)KKp * 4;
  constexpr int PW = KSP > 0 ? 4 / (IT * KSP) : (NK > 0 ? 1 : 4 / IT);
  const long long Rc = (long long)a.B * (g.InY / g.S) * (g.InX / g.S);
  unsigned tpc = (unsigned)((Rc + 15) / 16); tpc = (tpc + PW - 1) / PW * PW;
  const int blocks = (int)((long long)g.S * g.S * tpc / PW);
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dxs_kernel<IT, NK, KSP>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_dxs_kernel<IT, NK, KSP>), dim3(blocks, ity), dim3(256), lds, s, a, l, tpc);
  return hipGetLastError();
}
template <int IT> static hipError_t launchConvDxsC(const ConvArgs& a, int l, hipStream_t s) {
  const int nk = convPad4(convClassK(a.L[l])) / 4;
  if (nk == 36) {                                                  // 16 filters of 6 x 6, stride 2: 16 x 3 x 3 taps per class
    // thousands of tiles (batch 128 of 20 x 20 positions: 3200): a tile per wavefront -- a quarter of the workgroups, no cross-wave join
    const long long tiles = ((long long)a.B * (a.L[l].InY / a.L[l].S) * (a.L[l].InX / a.L[l].S) + 15) / 16 * a.L[l].S * a.L[l].S;
    if (tiles >= 2048) return launchConvDxsT<1, 36, 1>(a, l, IT, s);      // (RACER_atari step, round 4: 139.6 -> 138.0 us; two wavefronts per tile: in between)
    return launchConvDxsT<1, 36>(a, l, IT, s);
  }
  return launchConvDxsT<IT, 0>(a, l, 1, s);
}
template <int IT, int NK> static hipError_t launchConvDxT(const ConvArgs& a, int l, long long R, int ity, const DenseRide* ride, hipStream_t s) {
  size_t lds = convDxLds(a.L[l], IT);
  constexpr int PW = NK > 0 ? 1 : 4 / IT;       // (NK > 0: the four waves of a workgroup share one tile's reduction)
  const int blocks = (int)((R + 16 * PW - 1) / (16 * PW));
  DenseRide rd{}; int extra = 0;
  if (ride && ride->tile1 > ride->tile0) { rd = *ride; rd.own = blocks; extra = (rd.tile1 - rd.tile0 + ity - 1) / ity; if (lds < (size_t)DWW_LDS) lds = DWW_LDS; }
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dx_kernel<IT, NK>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_dx_kernel<IT, NK>), dim3(blocks + extra, ity), dim3(256), lds, s, a, l, rd);
  return hipGetLastError();
}
template <int IT> static hipError_t launchConvDxC(const ConvArgs& a, int l, long long R, const DenseRide* ride, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const int nk = convPad4(g.KnC * g.KnY * g.KnX) / 4;
  if (nk == 128) return launchConvDxT<1, 128>(a, l, R, IT, ride, s);      // 32 filters of 4 x 4
  if (nk == 144) return launchConvDxT<1, 144>(a, l, R, IT, ride, s);      // 16 of 6 x 6, 64 of 3 x 3
  return launchConvDxT<IT, 0>(a, l, R, 1, ride, s);
}
bool conv_dx_rides(const ConvGeo& g) { return !convStrided(g) && (g.InC + 15) / 16 <= 4; }      // the launches that take DenseRide tiles
hipError_t launch_conv_dx(const ConvArgs& a, int l, hipStream_t s, const DenseRide* ride) {
  const ConvGeo& g = a.L[l];
  const long long R = (long long)a.B * g.InY * g.InX;
  const int IT = (g.InC + 15) / 16;
  if (convStrided(g)) {
    if (ride && ride->tile1 > ride->tile0) return hipErrorInvalidValue;      // (conv_dx_rides)
    if (IT == 1) return launchConvDxsC<1>(a, l, s);
    if (IT == 2) return launchConvDxsC<2>(a, l, s);
    if (IT <= 4) return launchConvDxsC<4>(a, l, s);
    return hipErrorInvalidValue;
  }
  if (IT == 1) return launchConvDxC<1>(a, l, R, ride, s);
  if (IT == 2) return launchConvDxC<2>(a, l, R, ride, s);
  if (IT <= 4) return launchConvDxC<4>(a, l, R, ride, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// dW: partial filter gradients per (layer, 16 x 16 tile of [KnC][K], chunk of the batch x positions reduction)
// ---------------------------------------------------------------------------------------------------------------
constexpr int CONV_DW_MAXROWS = 2048;      // rows of one chunk (LDS tables)
constexpr size_t CONV_DW_LDS = 40 * 1024;  // (gather form: MAXROWS * 12 + 4 KB; staged form: conv_dw_staged_group keeps a workgroup's operands below this)
// ---- the same with BOTH operands staged in LDS (round 5; VERDICT r02 - r04: the gather form above requests each of its two operands
// per MFMA step with a 4-byte load of its own -- 16 cache lines per wavefront load for the deltas -- and stalls at issue 28 - 44 % of its
// cycles).  Behind the first layer a sample's input image and its deltas are a few KB, contiguous in memory: a workgroup copies the
// images of G rows and the deltas of one tile of 16 channels into LDS with 16-byte loads (flat copies, all in flight at once) and
// takes every MFMA operand from there.  A wavefront owns up to five of the K / 16 patch-element tiles and reads the delta operand
// once per step for all of them.  One partial [16][K] per (group of rows, channel tile) -> part[group][c][k], summed in group order by
// conv_reduce_adam_kernel: bit-deterministic.
int conv_dw_staged_group(const ConvGeo& g, int B) {      // rows per workgroup (0: the layer keeps the gather form)
  const long long inSize = (long long)g.InC * g.InY * g.InX;
  if ((inSize & 3) || (g.ldIn & 3) || (g.ldOut & 3) || (g.P * 16) % 4 || g.KnC % 16 || g.K % 16 || g.K / 16 > 20) return 0;
  int G = 1;
  while (2 * G <= 16 && 2 * G * g.P <= 128) G *= 2;      // reductions of about a hundred rows: 72 - 128 on the RACER_atari shape
  if (G > B) G = B;
  auto bytes = [&](int G_) { return ((long long)G_ * (inSize + 16 * (g.P | 1)) + 2 * (((long long)G_ * g.P + 3) & ~3)) * 4; };
  while (G > 1 && bytes(G) > (long long)CONV_DW_LDS) G /= 2;
  return bytes(G) <= (long long)CONV_DW_LDS ? G : 0;
}
__device__ __forceinline__ void convDwStaged(const ConvArgs& a, const ConvGeo& g, int local, unsigned char* smem) {
  const int tilesC = g.KnC >> 4, grp = local / tilesC, ct = local - grp * tilesC;
  const int G = g.dwG, b0 = grp * G, nb = min(G, a.B - b0);
  const int P = g.P, PP = P | 1, K = g.K, inSize = g.InC * g.InY * g.InX, R = nb * P, R4 = (R + 3) & ~3;
  float* sIn = reinterpret_cast<float*>(smem);                      // [G][inSize]
  float* sDl = sIn + (size_t)G * inSize;                             // [G][16][PP]   (odd pitch: the sixteen channels of an operand read fall into different banks)
  int* rIn = reinterpret_cast<int*>(sDl + (size_t)G * 16 * PP);      // [R4] patch origin of reduction row r in sIn
  int* rD = rIn + (((size_t)G * P + 3) & ~3);                        // [R4] its delta in sDl (channel 0 of the tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  {      // the rows' images: flat 16-byte copies (inSize floats at pitch ldIn), every load of a batch in flight at once
    const int q4 = inSize >> 2, nIn4 = nb * q4;
    constexpr int U = 6;
    for (int i0 = tid; i0 < nIn4; i0 += 256 * U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u;
        if (i < nIn4) { const int bl = i / q4, e = i - bl * q4; v[u] = reinterpret_cast<const f32x4*>(g.in + (size_t)(b0 + bl) * g.ldIn)[e]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u;
        if (i < nIn4) { const int bl = i / q4, e = i - bl * q4; reinterpret_cast<f32x4*>(sIn + (size_t)bl * inSize)[e] = v[u]; }
      }
    }
    // their deltas of this tile's 16 channels: 16 P consecutive floats per row -> [c][PP]
    const int nD = 16 * P, totD = nb * nD;
    for (int i = tid; i < totD; i += 256) {
      const int bl = i / nD, e = i - bl * nD, c = e / P, p = e - c * P;
      sDl[(bl * 16 + c) * PP + p] = g.D[(size_t)(b0 + bl) * g.ldOut + (size_t)ct * nD + e];
    }
  }
  for (int r = tid; r < R4; r += 256) {
    const int rr = r < R ? r : 0, bl = rr / P, p = rr - bl * P, oy = p / g.OpX, ox = p - oy * g.OpX;
    rIn[r] = bl * inSize + oy * g.S * g.InX + ox * g.S;
    rD[r] = bl * 16 * PP + p;
  }
  // this wavefront's patch-element tiles kt = wave, wave + 4, ...; the lane's element k = kt 16 + li of each
  constexpr int TP = 5;
  const int nKt = K >> 4, fsz = g.KnY * g.KnX;
  int ko[TP]; f32x4 acc[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int kt = min(wave + 4 * j, nKt - 1), k = kt * 16 + li, ic = k / fsz, f = k - ic * fsz, fy = f / g.KnX, fx = f - fy * g.KnX;
    ko[j] = ic * g.InY * g.InX + fy * g.InX + fx;
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nT = (nKt - wave + 3) >> 2;      // tiles of this wavefront (<= TP: conv_dw_staged_group)
  __syncthreads();
  const int nSteps = R4 >> 2;
  const float* dCol = sDl + li * PP;
  // software pipeline: the operands of step s + 1 (two table entries, then the delta and the TP patch elements they address) are read
  // while the MFMAs of step s run -- written as one loop the compiler leaves every step's LDS round trips in front of its MFMAs
  float dv, pv[TP];
  auto fetch = [&](int s, float& d, float (&pq)[TP]) {
    const int r = min(4 * s + lc, R4 - 1);
    const int oI = rIn[r], oD = rD[r];
    d = dCol[oD];
    d = r < R ? d : 0.f;
#pragma unroll
    for (int j = 0; j < TP; ++j) pq[j] = sIn[oI + ko[j]];
  };
  fetch(0, dv, pv);
  for (int s = 0; s < nSteps; ++s) {
    float dn, pn[TP];
    fetch(min(s + 1, nSteps - 1), dn, pn);
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      if (j < nT) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, pv[j], acc[j], 0, 0, 0);
    }
    dv = dn;
#pragma unroll
    for (int j = 0; j < TP; ++j) pv[j] = pn[j];
  }
  float* part = g.part + (size_t)grp * g.KnC * K + (size_t)(ct * 16 + lc * 4) * K + li;
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    if (j < nT) {
      const int kt = wave + 4 * j;
#pragma unroll
      for (int q = 0; q < 4; ++q) part[(size_t)q * K + kt * 16] = acc[j][q];
    }
  }
}
__device__ __forceinline__ void convDwBody(const ConvArgs& a, int bx, unsigned char* smem) {
  long long* sIn = reinterpret_cast<long long*>(smem);                       // [MAXROWS] offset of the patch origin of row r in the input array
  int* sD = reinterpret_cast<int*>(smem + (size_t)CONV_DW_MAXROWS * 8);      // [MAXROWS] b * ldOut + p
  float* red = reinterpret_cast<float*>(smem + (size_t)CONV_DW_MAXROWS * 12);  // [4][256]
  int l = 0;
  for (int i = 1; i < a.nL; ++i) if (bx >= a.L[i].dwBlock0) l = i;
  const ConvGeo g = a.L[l];
  if (g.dwG) { convDwStaged(a, g, bx - g.dwBlock0, smem); return; }
  const int K = g.K, P = g.P, tilesK = (K + 15) / 16, tilesC = (g.KnC + 15) / 16;
  const int local = bx - g.dwBlock0;
  const int chunk = local / (tilesK * tilesC), tile = local - chunk * tilesK * tilesC;
  const int ct = tile / tilesK, kt = tile - ct * tilesK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int R = a.B * P;
  const int r0 = chunk * g.chunkRows;
  const int nr = R - r0 < g.chunkRows ? R - r0 : g.chunkRows;
  for (int i = tid; i < nr; i += 256) {
    const unsigned r = (unsigned)(r0 + i);
    const int b = (int)(r / (unsigned)P), p = (int)(r - (unsigned)b * (unsigned)P), oy = p / g.OpX, ox = p - oy * g.OpX;
    sIn[i] = (long long)b * g.ldIn + (long long)oy * g.S * g.InX + ox * g.S;
    sD[i] = b * g.ldOut + p;
  }
  // this lane's patch element (column of the tile) and channel (row of the tile)
  const int k = kt * 16 + li, c = ct * 16 + li;
  int ko = 0;
  if (k < K) { const int ic = k / (g.KnY * g.KnX), f = k - ic * g.KnY * g.KnX, fy = f / g.KnX, fx = f - fy * g.KnX;
    ko = ic * g.InY * g.InX + fy * g.InX + fx; }
  const bool kOk = k < K, cOk = c < g.KnC;
  const int cOff = c * P;
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* in = g.in; const float* D = g.D;
  const int nGroups = (nr + 3) / 4;                      // groups of 4 rows = one MFMA step; wave w takes groups w, w+4, ...
  constexpr int UN = 16;                                 // operands of 16 steps in flight
  for (int g0 = wave; g0 < nGroups; g0 += 4 * UN) {
    float av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = 4 * (g0 + 4 * u) + lc;
      const bool rOk = i < nr && g0 + 4 * u < nGroups;
      av[u] = (rOk && cOk) ? D[sD[rOk ? i : 0] + cOff] : 0.f;
      bv[u] = (rOk && kOk) ? in[sIn[rOk ? i : 0] + ko] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lc * 4 + r) * 16 + li] = acc0[r] + acc1[r];
  __syncthreads();
  const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  const int oc = ct * 16 + (tid >> 4), okk = kt * 16 + (tid & 15);
  if (oc < g.KnC && okk < K) g.part[(size_t)chunk * g.KnC * K + (size_t)oc * K + okk] = v;
}
__global__ __launch_bounds__(256) void conv_dw_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  convDwBody(a, (int)blockIdx.x, smem);
}
hipError_t launch_conv_dw(const ConvArgs& a, int totalBlocks, hipStream_t s) {
  hipLaunchKernelGGL(conv_dw_kernel, dim3(totalBlocks), dim3(256), CONV_DW_LDS, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Row-block kernels for layers with a LARGE input image (the first layer of the Atari stacks: 84 x 84 x 4 -> k8 s4).
// The gather kernels above fetch every patch element with a 4-byte load of its own -- 64 loads per lane and tile in the
// forward pass, two gathers per MFMA step in the filter gradient: the vector-memory front end, not the matrix pipe, is
// their limit (19.6 and 31 us on the RACER_atari shape).  Here a workgroup owns rbRows output rows of ONE sample: the
// rbWin = (rbRows - 1) S + KnY input rows under them are contiguous per channel, so they are staged in LDS once with
// 16-byte loads (32 KB for 5 output rows of the 84 x 84 x 4 layer) and every patch element is an LDS read; the MFMA A
// operand -- the filter rows (forward) or the layer's deltas (filter gradient) -- is held in registers across the
// workgroup's tiles, so a step costs ONE LDS read.
//   forward:          Y[c][(b, p)] = sum_k K[c][k] patch_k(b, p)          tiles of 16 positions, waves take tiles 0, 4, ...
//   filter gradient:  dK[c][k]    = sum_p D[b][c][p] patch_k(b, p)        one partial [KnC][K] per (sample, row block), summed in
//                                                                         (sample, block) order by conv_reduce_adam_kernel
// Requirements (conv_row_block): InX a multiple of 4, K a multiple of 16, window + operands within 64 KB of LDS.
// ---------------------------------------------------------------------------------------------------------------
int conv_rows_atari_rb(const ConvGeo& g);
int conv_row_block(const ConvGeo& g, int* win) {
  if (const int rbA = conv_rows_atari_rb(g)) { if (win) *win = (rbA - 1) * g.S + g.KnY; return rbA; }      // (whole blocks: the kernels with this layer's geometry at compile time)
  if ((g.InX & 3) || (g.KnX & 3) || (g.K & 15) || g.K > 512 || g.KnC > 32 || (long long)g.InC * g.InY * g.InX < 4096) return 0;
  int best = 0; double bestEff = 0;
  for (int rb = 1; rb <= g.OpY; ++rb) {
    const int wr = (rb - 1) * g.S + g.KnY;
    const size_t lds = ((size_t)g.InC * wr * g.InX + (size_t)((g.KnC + 15) & ~15) * (g.K + 4) + g.K + 4 * 256) * 4;
    if (lds > 56 * 1024) break;
    const int tiles = (rb * g.OpX + 15) / 16;
    const double eff = (double)(rb * g.OpX) / (16.0 * ((tiles + 3) / 4 * 4));      // filled MFMA columns per round of four wavefronts
    if (eff > bestEff + 1e-9 || (eff > bestEff - 0.05 && rb > best)) { if (eff > bestEff) bestEff = eff; best = rb; }
  }
  if (best && win) *win = (best - 1) * g.S + g.KnY;
  return best;
}
// stage rows [iy0, iy0 + wr) of every input channel of sample row `in` into sIn[ic][wr][InX] (16-byte loads, all in flight per batch)
__device__ __forceinline__ void stageWindow(float* sIn, const float* in, const ConvGeo& g, int iy0, int wr, int wrValid) {
  const int rowF4 = g.InX >> 2, perCh = wrValid * rowF4, total = g.InC * perCh;
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  const int chStride4 = (g.InY * g.InX) >> 2, base4 = (iy0 * g.InX) >> 2, ldsCh4 = (wr * g.InX) >> 2;
  f32x4* dst = reinterpret_cast<f32x4*>(sIn);
  for (int q0 = 0; q0 < total; q0 += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = q0 + threadIdx.x + 256 * u; if (i < total) { const int ic = i / perCh, r = i - ic * perCh; v[u] = src[ic * chStride4 + base4 + r]; } }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = q0 + threadIdx.x + 256 * u; if (i < total) { const int ic = i / perCh, r = i - ic * perCh; dst[ic * ldsCh4 + r] = v[u]; } }
  }
}
// the same window straight from the replay (stack_gather_kernel's mapping): input channel ic = frame j = ic / C0 steps back (steps
// before the first repeat the first), channel c0 = ic % C0 of that state; standardised with the per-component mean and scale
// (slot / step of the row's state: loads of their own so that a caller can request them in front of a test that itself waits for a load)
__device__ __forceinline__ void replayRowOrigin(const ConvSource& src, int row, int B, long long* slot, int* t) {
  const int b = row < B ? row : src.nextSrc[row - B];
  *slot = src.slot[b] + (row < B ? 0 : 1);
  *t = src.t[b] + (row < B ? 0 : 1);
}
__device__ __forceinline__ void stageWindowReplayAt(float* sIn, const ConvSource& src, long long slot, int t, const ConvGeo& g, int iy0, int wr, int wrValid);
__device__ __forceinline__ void stageWindowReplay(float* sIn, const ConvSource& src, int row, int B, const ConvGeo& g, int iy0, int wr, int wrValid) {
  long long slot; int t;
  replayRowOrigin(src, row, B, &slot, &t);
  stageWindowReplayAt(sIn, src, slot, t, g, iy0, wr, wrValid);
}
__device__ __forceinline__ void stageWindowReplayAt(float* sIn, const ConvSource& src, long long slot, int t, const ConvGeo& g, int iy0, int wr, int wrValid) {
  const int rowF4 = g.InX >> 2, perCh = wrValid * rowF4, total = g.InC * perCh;
  const int C0 = g.InC / (1 + src.nApp), chStride4 = (g.InY * g.InX) >> 2, base4 = (iy0 * g.InX) >> 2, ldsCh4 = (wr * g.InX) >> 2;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(src.mean); const f32x4* s4 = reinterpret_cast<const f32x4*>(src.scale);
  f32x4* dst = reinterpret_cast<f32x4*>(sIn);
#ifndef CONV_WU
#define CONV_WU 4
#endif
  constexpr int WU = CONV_WU;      // (8 -- the whole 84 x 84 x 4 window of five output rows in one batch -- costs a wavefront per SIMD: 143.6 -> 152.5 us per step)
  for (int q0 = 0; q0 < total; q0 += 256 * WU) {
    f32x4 v[WU], mu[WU], sc[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = q0 + threadIdx.x + 256 * u;
      if (i < total) {
        const int ic = i / perCh, r = i - ic * perCh, j = ic / C0, c0 = ic - j * C0, back = j < t ? j : t;
        const int off4 = c0 * chStride4 + base4 + r;
        v[u] = reinterpret_cast<const f32x4*>(src.S + (size_t)(slot - back) * src.dS)[off4]; mu[u] = m4[off4]; sc[u] = s4[off4];
      }
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = q0 + threadIdx.x + 256 * u;
      if (i < total) {
        const int ic = i / perCh, r = i - ic * perCh;
        f32x4 o; o[0] = (v[u][0] - mu[u][0]) * sc[u][0]; o[1] = (v[u][1] - mu[u][1]) * sc[u][1]; o[2] = (v[u][2] - mu[u][2]) * sc[u][2]; o[3] = (v[u][3] - mu[u][3]) * sc[u][3];
        dst[ic * ldsCh4 + r] = o;
      }
    }
  }
}
template <int NK, int CT, int KNY, int KNX>       // MFMA steps per tile (K / 4), channel tiles of 16, filter size (compile time: no index divisions in the loop)
__global__ __launch_bounds__(256) void conv_fwd_rows_kernel(ConvArgs a, int l) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo g = a.L[l];
  const int row = blockIdx.y;
  // (the row's replay slot is requested beside the row count, not behind the test on it: one dependent round trip less in front of the window's loads)
  const int nRowsNow = a.sc->nRows[a.parity];
  long long rowSlot = 0; int rowT = 0;
  if (a.src.on && row < a.B) replayRowOrigin(a.src, row, a.B, &rowSlot, &rowT);      // (minibatch rows: entries the sampler always writes)
  if (row >= nRowsNow) return;
  if (a.src.on && row >= a.B) replayRowOrigin(a.src, row, a.B, &rowSlot, &rowT);     // (next-state rows: their map is valid below the row count only)
  const int rb = blockIdx.x, RB = g.rbRows, WR = g.rbWin, P = g.P, K = g.K, ldK = convPad4(K) + 4;
  const int oy0 = rb * RB, nOy = min(RB, g.OpY - oy0), iy0 = oy0 * g.S, wrValid = min(WR, g.InY - iy0);
  float* sIn = reinterpret_cast<float*>(smem);                       // [InC][WR][InX]
  float* Ws = sIn + (size_t)g.InC * WR * g.InX;                      // [CT 16][ldK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  // (the filter rows' loads are issued in front of the window's and stored behind them: one exposed latency, not two)
  constexpr int WQ = (CT * 16 * (4 * NK + 4) / 4 + 255) / 256;
  f32x4 wv[WQ];
  {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(g.Wf); const int n4 = (CT * 16 * ldK) >> 2;
#pragma unroll
    for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; wv[q] = i < n4 ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  if (a.src.on) stageWindowReplayAt(sIn, a.src, rowSlot, rowT, g, iy0, WR, wrValid);
  else stageWindow(sIn, g.in + (long long)row * g.ldIn, g, iy0, WR, wrValid);
  {
    f32x4* d4 = reinterpret_cast<f32x4*>(Ws); const int n4 = (CT * 16 * ldK) >> 2;
#pragma unroll
    for (int q = 0; q < WQ; ++q) { const int i = tid + 256 * q; if (i < n4) d4[i] = wv[q]; }
  }
  // window-relative offset of patch element k = 4 s + lc: KnX is a multiple of 4, so the four elements of a step lie side by
  // side in one filter row -- offset(4 s) is uniform (scalar registers), the lane adds lc
  constexpr int fsz = KNY * KNX;
  __syncthreads();
  float av[CT][NK];                                                  // this lane's filter row (channel li of tile ct), elements 4 s + lc
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int s = 0; s < NK; ++s) av[ct][s] = Ws[(ct * 16 + li) * ldK + 4 * s + lc];
  }
  const int nPos = nOy * g.OpX, nTiles = (nPos + 15) >> 4;
  const float* Bl = a.W + g.indB;
  for (int t = wave; t < nTiles; t += 4) {
    const int pl = t * 16 + li; const bool ok = pl < nPos;
    const int plc = ok ? pl : 0, oyl = plc / g.OpX, ox = plc - oyl * g.OpX;
    const float* pIn = sIn + oyl * g.S * g.InX + ox * g.S + lc;
    f32x4 acc[CT][2];
    float bq[CT][4];      // the biases of this lane's outputs, requested in front of the tile's MFMA chain
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { acc[ct][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = acc[ct][0];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int ch = ct * 16 + lc * 4 + q; bq[ct][q] = (ok && ch < g.KnC) ? Bl[(size_t)ch * P + oy0 * g.OpX + pl] : 0.f; } }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      constexpr int dummy = 0; (void)dummy;
      const int k0 = 4 * s, ic = k0 / fsz, f = k0 - ic * fsz, fy = f / KNX, fx = f - fy * KNX;      // (constants after unrolling)
      const float bv = pIn[(ic * WR + fy) * g.InX + fx];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct][s], bv, acc[ct][s & 1], 0, 0, 0);
    }
    if (ok) {
      const int pp = oy0 * g.OpX + pl;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = ct * 16 + lc * 4 + q;
          if (ch < g.KnC) {
            const float x = (acc[ct][0][q] + acc[ct][1][q]) + bq[ct][q];
            const size_t o = (size_t)row * g.ldOut + (size_t)ch * P + pp;
            g.X[o] = x; g.Y[o] = softsignEval(x);
          }
        }
      }
    }
  }
}
static size_t convRowsFwdLds(const ConvGeo& g) { return ((size_t)g.InC * g.rbWin * g.InX + (size_t)((g.KnC + 15) & ~15) * (convPad4(g.K) + 4)) * 4; }
template <int NK, int CT> static hipError_t launchFwdRowsT(const ConvArgs& a, int l, int maxRows, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const size_t lds = convRowsFwdLds(g);
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_fwd_rows_kernel<NK, CT, 8, 8>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_fwd_rows_kernel<NK, CT, 8, 8>), dim3(g.rbCount, maxRows), dim3(256), lds, s, a, l);
  return hipGetLastError();
}
hipError_t launch_conv_forward_rows_atari(const ConvArgs& a, int maxRows, hipStream_t s);
hipError_t launch_conv_forward_rows(const ConvArgs& a, int l, int maxRows, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  if (l == 0 && g.rbKind == 1) return launch_conv_forward_rows_atari(a, maxRows, s);
  const int nk = g.K / 4, ct = (g.KnC + 15) / 16;
  if (nk == 64 && ct == 1) return launchFwdRowsT<64, 1>(a, l, maxRows, s);      // 4 x 8 x 8 patches, <= 16 filters (RACER_atari.json)
  if (nk == 64 && ct == 2) return launchFwdRowsT<64, 2>(a, l, maxRows, s);      // ... 32 filters (the Atari paper's first layer)
  return hipErrorInvalidValue;
}
bool conv_rows_ok(const ConvGeo& g) { return g.K == 256 && g.KnY == 8 && g.KnX == 8 && g.KnC <= 32; }      // the instantiated shape: 4 x 8 x 8 patches


// ---- the same two kernels for THE first layer of RACER_atari.json (84 x 84 x (1 + 3 appended frames) -> 8 filters of 8 x 8, stride 4;
// windows read from the replay) with the geometry at compile time (round 5).  The any-geometry kernels above spend most of their
// instructions on index arithmetic (SQ_INSTS_VALU: 2.5 M wavefront instructions per launch of the forward kernel, 0.3 M of them
// operand reads); these launches are bound by what their wavefronts ISSUE, so the arithmetic is removed:
//   * the window of a row block (24 image rows x 4 frames) is one flat range of 504 float4 per frame: a thread owns two positions of
//     that range, loads mean and scale ONCE and the pixel of each of the four frames (6 requests instead of 12, no divisions);
//   * the reduction index is permuted inside groups of 16 (lane group lc takes k = 16 g + 4 lc + j in sub-step j, for both operands):
//     the four patch elements of a lane's sub-steps are four neighbouring pixels -> ONE 16-byte LDS read per four MFMAs, and the
//     filter rows come as 16-byte loads straight into registers (forward);
//   * filter gradient: every LDS address is a per-lane base + a compile-time offset (a row of 20 positions = five groups of four).
template <int RB_> struct Atari0 {
  static constexpr int INC = 4, IN = 84, KN = 8, S = 4, KNC = 8, OP = 20, K = 256, P = 400, RB = RB_, WR = (RB_ - 1) * 4 + 8, RBC = OP / RB_;
  static constexpr int ROW4 = IN / 4, WIN4 = WR * ROW4, DS4 = IN * IN / 4;      // float4 per image row / window of one frame / frame
  static constexpr int NPOS = RB * OP, NTILE = (NPOS + 15) / 16, NU = (WIN4 + 255) / 256;
  static constexpr int LD_D = NPOS;                                             // pitch of the staged deltas [16][LD_D]
  static constexpr size_t LDS_FWD = (size_t)INC * WR * IN * 4;
  static constexpr size_t LDS_DW = (size_t)(INC * WR * IN + 16 * LD_D) * 4;
  static_assert(OP % RB_ == 0 && NPOS % 4 == 0 && KNC * (NPOS / 4) <= 256, "whole row blocks");
};
int conv_rows_atari_rb(const ConvGeo& g) {      // rows per block of this layer when it is the first layer of RACER_atari.json (0: another layer)
  if (!(g.InC == 4 && g.InY == 84 && g.InX == 84 && g.KnY == 8 && g.KnX == 8 && g.S == 4 && g.KnC == 8 && g.OpY == 20 && g.OpX == 20)) return 0;
  return 4;      // (RACER_atari step: 97.3 us at 4 -- five full position tiles per block --, 98.3 at 5, 103.5 at 2)
}
bool conv_rows_atari(const ConvGeo& g, const ConvSource& src) {
  const int rb = conv_rows_atari_rb(g);
  return rb && src.on && src.nApp == 3 && src.dS == 84 * 84 && (g.rbRows == 2 || g.rbRows == 4 || g.rbRows == 5) && g.rbWin == (g.rbRows - 1) * 4 + 8 && g.rbCount == 20 / g.rbRows;
}
// the window of row block rb of the state at (slot, t): sIn[frame][WR rows][84], standardised (Episode.h:172-183: frame j = j steps back, the
// first state repeated in front of the episode's start)
template <class A0>
__device__ __forceinline__ void atari0StageWindow(float* __restrict__ sIn, const ConvSource& src, long long slot, int t, int rb) {
  const f32x4* m4 = reinterpret_cast<const f32x4*>(src.mean) + rb * (A0::RB * A0::S) * A0::ROW4;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src.scale) + rb * (A0::RB * A0::S) * A0::ROW4;
  const f32x4* S4 = reinterpret_cast<const f32x4*>(src.S) + rb * (A0::RB * A0::S) * A0::ROW4;
  const int tid = threadIdx.x;
  f32x4 mu[A0::NU], sc[A0::NU], v[A0::NU][A0::INC];
#pragma unroll
  for (int u = 0; u < A0::NU; ++u) {
    const int w = tid + 256 * u;
    if (w < A0::WIN4) {
      mu[u] = m4[w]; sc[u] = s4[w];
#pragma unroll
      for (int j = 0; j < A0::INC; ++j) { const int back = j < t ? j : t; v[u][j] = S4[(size_t)(slot - back) * A0::DS4 + w]; }
    }
  }
#pragma unroll
  for (int u = 0; u < A0::NU; ++u) {
    const int w = tid + 256 * u;
    if (w < A0::WIN4) {
#pragma unroll
      for (int j = 0; j < A0::INC; ++j) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[u][j][e] - mu[u][e]) * sc[u][e];
        reinterpret_cast<f32x4*>(sIn)[j * A0::WIN4 + w] = o;
      }
    }
  }
}
template <int RB>
__global__ __launch_bounds__(256) void conv_fwd_rows_atari_kernel(ConvArgs a) {
  using A0 = Atari0<RB>;
  constexpr int K = A0::K, IN = A0::IN, S = A0::S, OP = A0::OP, P = A0::P, WR = A0::WR, NPOS = A0::NPOS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo& g = a.L[0];
  const int row = blockIdx.y, rb = blockIdx.x;
  const int nRowsNow = a.sc->nRows[a.parity];
  long long rowSlot = 0; int rowT = 0;
  if (row < a.B) replayRowOrigin(a.src, row, a.B, &rowSlot, &rowT);      // (minibatch rows: requested beside the row count, not behind the test on it)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  // this lane's filter row (channel li; rows 8 .. 15 of the padded layout are zero), elements 16 G + 4 lc .. + 3 of every group G
  f32x4 av[K / 16];
  {
    const f32x4* wf = reinterpret_cast<const f32x4*>(g.Wf + (size_t)li * (K + 4)) + lc;
#pragma unroll
    for (int G = 0; G < K / 16; ++G) av[G] = wf[4 * G];
  }
  if (row >= nRowsNow) return;
  if (row >= a.B) replayRowOrigin(a.src, row, a.B, &rowSlot, &rowT);
  float* sIn = reinterpret_cast<float*>(smem);
  atari0StageWindow<A0>(sIn, a.src, rowSlot, rowT, rb);
  __syncthreads();
  const float* Bl = a.W + g.indB + rb * NPOS;
  float* Xr = g.X + (size_t)row * g.ldOut + rb * NPOS; float* Yr = g.Y + (size_t)row * g.ldOut + rb * NPOS;
  // patch element k = 16 G + 4 lc + j: frame G / 4, filter row 2 (G % 4) + lc / 2, filter columns 4 (lc % 2) + j
  const int laneOff = (lc >> 1) * IN + (lc & 1) * 4;
  for (int tl = wave; tl < A0::NTILE; tl += 4) {
    const int pl = tl * 16 + li; const bool ok = pl < NPOS;
    const int pc = ok ? pl : 0, oyl = pc / OP, ox = pc - oyl * OP;
    float bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = (ok && lc < 2) ? Bl[(4 * lc + q) * P + pl] : 0.f;
    const f32x4* pIn = reinterpret_cast<const f32x4*>(sIn + oyl * S * IN + ox * S + laneOff);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int G = 0; G < K / 16; ++G) {
      const f32x4 bv = pIn[((G >> 2) * WR * IN + 2 * (G & 3) * IN) / 4];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[G][0], bv[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[G][1], bv[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[G][2], bv[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[G][3], bv[3], acc1, 0, 0, 0);
    }
    if (ok && lc < 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = (acc0[q] + acc1[q]) + bq[q];
        Xr[(4 * lc + q) * P + pl] = x; Yr[(4 * lc + q) * P + pl] = softsignEval(x);
      }
    }
  }
}
template <int RB> static hipError_t launchFwdRowsAtariT(const ConvArgs& a, int maxRows, hipStream_t s) {
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_fwd_rows_atari_kernel<RB>), Atari0<RB>::LDS_FWD);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_fwd_rows_atari_kernel<RB>, dim3(Atari0<RB>::RBC, maxRows), dim3(256), Atari0<RB>::LDS_FWD, s, a);
  return hipGetLastError();
}
hipError_t launch_conv_forward_rows_atari(const ConvArgs& a, int maxRows, hipStream_t s) {
  const int rb = a.L[0].rbRows;
  return rb == 5 ? launchFwdRowsAtariT<5>(a, maxRows, s) : (rb == 4 ? launchFwdRowsAtariT<4>(a, maxRows, s) : launchFwdRowsAtariT<2>(a, maxRows, s));
}
// filter gradient of (row block rb, row b): part[(b RBC + rb)][c][k], c < 8
template <int RB>
__device__ __forceinline__ void convDwRowsAtariT(const ConvArgs& a, int rb, int b, unsigned char* smem) {
  using A0 = Atari0<RB>;
  constexpr int K = A0::K, IN = A0::IN, S = A0::S, P = A0::P, WR = A0::WR, NPOS = A0::NPOS, LD_D = A0::LD_D, KNC = A0::KNC;
  const ConvGeo& g = a.L[0];
  float* sIn = reinterpret_cast<float*>(smem);
  float* sD = sIn + A0::INC * WR * IN;                   // [16][LD_D]: deltas of the block's positions, channels 8 .. 15 zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  long long slot; int t;
  replayRowOrigin(a.src, b, a.B, &slot, &t);
  f32x4 dv = {0.f, 0.f, 0.f, 0.f};
  const int dc = tid / (NPOS / 4), dq = tid - dc * (NPOS / 4);      // NPOS / 4 float4 per channel
  if (tid < KNC * (NPOS / 4)) dv = reinterpret_cast<const f32x4*>(g.D + (size_t)b * g.ldOut + (size_t)dc * P + rb * NPOS)[dq];
  atari0StageWindow<A0>(sIn, a.src, slot, t, rb);
  if (tid < KNC * (NPOS / 4)) reinterpret_cast<f32x4*>(sD + dc * LD_D)[dq] = dv;
  for (int i = tid; i < 8 * LD_D; i += 256) sD[8 * LD_D + i] = 0.f;
  __syncthreads();
  // this wavefront's patch-element tiles kt = wave, wave + 4, wave + 8, wave + 12; the lane's element k = kt 16 + li of each:
  // frame k / 64, filter row (k % 64) / 8, column k % 8; reduction row r = 4 s + lc = position (s / 5, 4 (s % 5) + lc) of the block
  const float* pB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = (wave + 4 * j) * 16 + li; pB[j] = sIn + (k >> 6) * WR * IN + ((k >> 3) & 7) * IN + (k & 7) + lc * S; }
  const float* pA = sD + li * LD_D + lc;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < NPOS / 4; ++s) {
    const float d = pA[4 * s];
    const int off = (s / 5) * S * IN + (s % 5) * 4 * S;      // (a constant after unrolling)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, pB[j][off], acc[j], 0, 0, 0);
  }
  if (lc < 2) {
    float* part = g.part + ((size_t)b * A0::RBC + rb) * (size_t)KNC * K + (size_t)(4 * lc) * K + li;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[(size_t)q * K + (wave + 4 * j) * 16] = acc[j][q];
  }
}
__device__ __forceinline__ void convDwRowsAtariBody(const ConvArgs& a, int bx, unsigned char* smem) {      // bx = b rbCount + rb
  const int rbr = a.L[0].rbRows;
  if (rbr == 5) convDwRowsAtariT<5>(a, bx & 3, bx >> 2, smem);
  else if (rbr == 4) convDwRowsAtariT<4>(a, bx % 5, bx / 5, smem);
  else convDwRowsAtariT<2>(a, bx % 10, bx / 10, smem);
}

// filter gradient of such a layer: partial [KnC][K] of (sample b, row block rb) -> part[(b nRB + rb)][c][k]
template <int CT>
__device__ __forceinline__ void convDwRowsBody(const ConvArgs& a, int l, int rb, int b, unsigned char* smem) {
  const ConvGeo g = a.L[l];
  const int RB = g.rbRows, WR = g.rbWin, P = g.P, K = g.K;
  const int oy0 = rb * RB, nOy = min(RB, g.OpY - oy0), iy0 = oy0 * g.S, wrValid = min(WR, g.InY - iy0);
  const int nPos = nOy * g.OpX, nPos4 = (nPos + 3) & ~3, ldD = RB * g.OpX + 4;
  float* sIn = reinterpret_cast<float*>(smem);                       // [InC][WR][InX]
  float* sD = sIn + (size_t)g.InC * WR * g.InX;                      // [CT 16][ldD]   deltas of this block's positions, zero padded
  int* sPos = reinterpret_cast<int*>(sD + (size_t)CT * 16 * ldD);    // [RB OpX + 4]    window offset of the patch origin of position r
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  // (the first deltas are requested in front of the window's pieces and stored behind them: their latency hides under the window's)
  constexpr int DQ = 8 * CT;
  float dreg[DQ];
  const int nD = CT * 16 * ldD;
#pragma unroll
  for (int q = 0; q < DQ; ++q) {
    const int i = tid + 256 * q, c = i / ldD, r = i - c * ldD;
    dreg[q] = (i < nD && c < g.KnC && r < nPos) ? g.D[(size_t)b * g.ldOut + (size_t)c * P + oy0 * g.OpX + r] : 0.f;
  }
  if (a.src.on) stageWindowReplay(sIn, a.src, b, a.B, g, iy0, WR, wrValid);
  else stageWindow(sIn, g.in + (long long)b * g.ldIn, g, iy0, WR, wrValid);
#pragma unroll
  for (int q = 0; q < DQ; ++q) { const int i = tid + 256 * q; if (i < nD) sD[i] = dreg[q]; }
  for (int i = tid + 256 * DQ; i < nD; i += 256) {
    const int c = i / ldD, r = i - c * ldD;
    sD[i] = (c < g.KnC && r < nPos) ? g.D[(size_t)b * g.ldOut + (size_t)c * P + oy0 * g.OpX + r] : 0.f;
  }
  for (int r = tid; r < nPos4; r += 256) { const int rr = r < nPos ? r : 0, oyl = rr / g.OpX, ox = rr - oyl * g.OpX; sPos[r] = oyl * g.S * g.InX + ox * g.S; }
  __syncthreads();
  const int nSteps = nPos4 >> 2;                                      // MFMA steps: four positions each
  // this wave's patch-element tiles: kt = wave, wave + 4, ... (K / 16 tiles); the lane's element k = kt 16 + li
  const int fsz = g.KnY * g.KnX, nKt = K >> 4;
  float* part = g.part + ((size_t)b * g.rbCount + rb) * (size_t)g.KnC * K;
  for (int kt0 = wave; kt0 < nKt; kt0 += 16) {                       // four tiles per pass: the delta operand is read once for all of them
    int ko[4]; f32x4 acc[CT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = min(kt0 + 4 * j, nKt - 1) * 16 + li, ic = k / fsz, f = k - ic * fsz, fy = f / g.KnX, fx = f - fy * g.KnX;
      ko[j] = (ic * WR + fy) * g.InX + fx;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s = 0; s < nSteps; ++s) {
      const int r = 4 * s + lc, po = sPos[r];
      float dv[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dv[ct] = sD[(ct * 16 + li) * ldD + r];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = sIn[po + ko[j]];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[ct], pv, acc[ct][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kt = kt0 + 4 * j;
      if (kt < nKt) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int c = ct * 16 + lc * 4 + q; if (c < g.KnC) part[(size_t)c * K + kt * 16 + li] = acc[ct][j][q]; }
        }
      }
    }
  }
}
template <int CT>
__global__ __launch_bounds__(256) void conv_dw_rows_kernel(ConvArgs a, int l) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (CT == 1 && a.L[l].rbKind == 1) convDwRowsAtariBody(a, (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, smem);
  else convDwRowsBody<CT>(a, l, (int)blockIdx.x, (int)blockIdx.y, smem);
}
// both filter-gradient launches as one (they depend on the deltas only): the first nRowBlocks workgroups take (sample, row block)
// pairs of layer l, the others the (layer, tile, chunk) problems of conv_dw_kernel
template <int CT>
__global__ __launch_bounds__(256) void conv_dw_all_kernel(ConvArgs a, int l, int nRowBlocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bx = (int)blockIdx.x;
  if (bx < nRowBlocks) {
    if (CT == 1 && a.L[l].rbKind == 1) convDwRowsAtariBody(a, bx, smem);
    else { const int rbCount = a.L[l].rbCount; convDwRowsBody<CT>(a, l, bx % rbCount, bx / rbCount, smem); }
  }
  else convDwBody(a, bx - nRowBlocks, smem);
}
// ... and, behind them, the tiles of the dense layers' weight gradients (dw_wide_dev.h: one workgroup per 16 x 16 tile, operands
// straight from memory, Adam in the epilogue; minibatches of at most 128 rows) with that launch's far-policy count + beta rider in front: three independent latency-bound families of
// workgroups share the chip instead of queueing as two launches (Layer_Conv2D.h:88-139 and Layers.h:164-187 need the deltas only)
template <int CT>
__global__ __launch_bounds__(256) void conv_dw_dense_kernel(ConvArgs a, int l, int nRowBlocks, int nConvDw, const GemmProblem* __restrict__ probs, int nProbs,
                                                            int nTiles, AdamHyper hyp, ExtraArgs extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int bx = (int)blockIdx.x;
#ifdef HL_CONVT_STAMPS
  // development (tools/convt_stamps.py dw): start of the launch's first workgroup, last end per family of workgroups (100 MHz clock)
  auto fin = [&](int fam) { __syncthreads(); if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(a.sc->dbgT) + 20 + fam, (unsigned long long)wall_clock64()); };
  if (blockIdx.x == 0 && threadIdx.x == 0) a.sc->dbgT[19] = wall_clock64();
#else
  auto fin = [&](int) {};
#endif
  if (extra.role) { if (bx == 0) { if (extra.role == 3) farBetaPhase(extra.post, smem); fin(0); return; } --bx; }
  if (bx < nRowBlocks) {
    if (CT == 1 && a.L[l].rbKind == 1) convDwRowsAtariBody(a, bx, smem);
    else { const int rbCount = a.L[l].rbCount; convDwRowsBody<CT>(a, l, bx % rbCount, bx / rbCount, smem); }
    fin(1); return;
  }
  bx -= nRowBlocks;
  if (bx < nConvDw) { convDwBody(a, bx, smem); fin(2); return; }
  dwWideBody<16>(probs, nProbs, nTiles, 1, nullptr, nullptr, a.sc, hyp, bx - nConvDw, smem);      // (B <= 128: launch_conv_dw_dense)
  fin(3);
}
static size_t convRowsDwLds(const ConvGeo& g) {
  const int ct = (g.KnC + 15) / 16, ldD = g.rbRows * g.OpX + 4;
  return ((size_t)g.InC * g.rbWin * g.InX + (size_t)ct * 16 * ldD + (size_t)g.rbRows * g.OpX + 4) * 4;
}
hipError_t launch_conv_dw_rows(const ConvArgs& a, int l, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const size_t lds = convRowsDwLds(g);
  const int ct = (g.KnC + 15) / 16;
  if (ct == 1) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_rows_kernel<1>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_rows_kernel<1>, dim3(g.rbCount, a.B), dim3(256), lds, s, a, l);
  } else if (ct == 2) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_rows_kernel<2>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_rows_kernel<2>, dim3(g.rbCount, a.B), dim3(256), lds, s, a, l);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

// sum of the chunk partials (+ Adam): 16 filter weights x 16 slices of the chunk range per workgroup -- a slice is summed in chunk
// order with eight partials in flight, the sixteen slice sums are added in slice order (fixed association: bit-deterministic);
// the row-block kernels leave one partial per (sample, row block), i.e. hundreds per weight
__global__ __launch_bounds__(256) void conv_reduce_adam_kernel(ConvArgs a, AdamHyper hyp, int fuseAdam) {
  __shared__ float sp[16][16];
  const int wl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  long long i = (long long)blockIdx.x * 16 + wl;
  int l = 0;
  for (; l < a.nL; ++l) { const long long n = (long long)a.L[l].KnC * a.L[l].K; if (i < n) break; i -= n; }
  const bool valid = l < a.nL;
  const ConvGeo& g = a.L[valid ? l : 0];
  const size_t n = (size_t)g.KnC * g.K;
  float s = 0.f;
  if (valid) {
    const int per = (g.nChunks + 15) / 16, c0s = sl * per, c1s = min(g.nChunks, c0s + per);
    for (int c0 = c0s; c0 < c1s; c0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = c0 + u < c1s ? g.part[(size_t)(c0 + u) * n + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  sp[sl][wl] = s;
  __syncthreads();
  if (sl != 0 || !valid) return;
  s = sp[0][wl];
#pragma unroll
  for (int q = 1; q < 16; ++q) s += sp[q][wl];
  a.G[g.indW + i] = s;
  if (fuseAdam) {
    AdamCoef c; c.eta = a.sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac;
    float w = a.Wrw[g.indW + i], m1 = a.M1[g.indW + i], m2 = a.M2[g.indW + i];
    adamStep(c, s, w, m1, m2);
    a.Wrw[g.indW + i] = w; a.M1[g.indW + i] = m1; a.M2[g.indW + i] = m2;
    // the new weight goes straight into the kernels' LDS layouts (what conv_prep_kernel would rebuild before the next forward pass)
    const int cc = (int)(i / g.K), k = (int)(i - (long long)cc * g.K);
    g.Wf[(size_t)cc * (convPad4(g.K) + 4) + k] = w;
    if (l > 0) {
      const int fsz = g.KnY * g.KnX, ic = k / fsz, f = k - ic * fsz;
      if (convStrided(g)) {
        const int S = g.S, TY = g.KnY / S, TX = g.KnX / S, ld = convPad4(g.KnC * TY * TX) + 4, rows = (g.InC + 15) & ~15;
        const int fy = f / g.KnX, fx = f - fy * g.KnX, cls = (fy % S) * S + (fx % S), kc = cc * TY * TX + (fy / S) * TX + (fx / S);
        g.Wx[((size_t)cls * rows + ic) * ld + kc] = w;
      } else {
        g.Wx[(size_t)ic * (convPad4(g.KnC * fsz) + 4) + cc * fsz + f] = w;
      }
    }
  }
}
hipError_t launch_conv_reduce_adam(const ConvArgs& a, const AdamHyper& hyp, int fuseAdam, hipStream_t s) {
  long long n = 0;
  for (int l = 0; l < a.nL; ++l) n += (long long)a.L[l].KnC * a.L[l].K;
  hipLaunchKernelGGL(conv_reduce_adam_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, a, hyp, fuseAdam);
  return hipGetLastError();
}

hipError_t launch_conv_dw_all(const ConvArgs& a, int l, int dwBlocks, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  size_t lds = convRowsDwLds(g); if (lds < CONV_DW_LDS) lds = CONV_DW_LDS;
  const int ct = (g.KnC + 15) / 16, nRowBlocks = g.rbCount * a.B;
  if (ct == 1) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_all_kernel<1>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_all_kernel<1>, dim3(nRowBlocks + dwBlocks), dim3(256), lds, s, a, l, nRowBlocks);
  } else if (ct == 2) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_all_kernel<2>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_all_kernel<2>, dim3(nRowBlocks + dwBlocks), dim3(256), lds, s, a, l, nRowBlocks);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_conv_dw_dense(const ConvArgs& a, int l, int dwBlocks, const GemmProblem* dProbs, int nProbs, int nTiles, const AdamHyper& hyp,
                                const ExtraArgs* extra, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  size_t lds = convRowsDwLds(g); if (lds < CONV_DW_LDS) lds = CONV_DW_LDS;
  if (lds < (size_t)DWW_LDS) lds = DWW_LDS;
  ExtraArgs ex{}; if (extra) ex = *extra;
  ex.helpers = 0;
  if (ex.role && lds < (size_t)TAIL_LDS_BYTES) lds = TAIL_LDS_BYTES;
  if (hyp.push.on || (ex.role && ex.role != 3)) return hipErrorInvalidValue;      // (replicas pushing tiles into peer windows keep the common launch; the one rider served: far-policy count + beta)
  const int ct = (g.KnC + 15) / 16, nRowBlocks = g.rbCount * a.B;
  const unsigned grid = (unsigned)((ex.role ? 1 : 0) + nRowBlocks + dwBlocks + ((nTiles + 7) / 8) * 8);
  if (ct == 1) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_dense_kernel<1>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_dense_kernel<1>, dim3(grid), dim3(256), lds, s, a, l, nRowBlocks, dwBlocks, dProbs, nProbs, nTiles, hyp, ex);
  } else if (ct == 2) {
    hipError_t e = ensureDynLds(reinterpret_cast<const void*>(conv_dw_dense_kernel<2>), lds); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_dw_dense_kernel<2>, dim3(grid), dim3(256), lds, s, a, l, nRowBlocks, dwBlocks, dProbs, nProbs, nTiles, hyp, ex);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace hl
