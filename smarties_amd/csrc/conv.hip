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

namespace hl {

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
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < dIn; idx += gridDim.x * 256) {
    const int j = idx / dS, i = idx - j * dS;
    const int back = j < t ? j : t;                           // steps before the first repeat the first
    a.X0[(size_t)row * a.ldX0 + idx] = (a.rp.S[(size_t)(slot - back) * dS + i] - a.rp.stMean[i]) * a.rp.stScale[i];
  }
}
hipError_t launch_stack_gather(const StackGatherArgs& a, int maxRows, hipStream_t s) {
  const int dIn = a.dS * (1 + a.nApp);
  int bx = (dIn + 255) / 256; if (bx > 64) bx = 64;
  hipLaunchKernelGGL(stack_gather_kernel, dim3(bx, maxRows), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
constexpr int CONV_PT = 2;     // 16-position tiles per wavefront

__host__ __device__ inline int convPad4(int k) { return (k + 3) & ~3; }
__host__ __device__ inline size_t convFwdLds(const ConvGeo& g, int CT) {
  const int Kp = convPad4(g.K);
  return (size_t)CT * 16 * (Kp + 4) * 4 + (size_t)Kp * 4;
}

template <int CT>     // 16-channel tiles
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a, int l) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo g = a.L[l];
  const int K = g.K, Kp = convPad4(K), ldK = Kp + 4, P = g.P;
  float* Ws = reinterpret_cast<float*>(smem);                         // [CT*16][ldK]
  int* kOff = reinterpret_cast<int*>(Ws + (size_t)CT * 16 * ldK);     // [Kp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int nRows = a.sc->nRows[a.parity];
  const long long R = (long long)nRows * P;
  const long long tile0 = ((long long)blockIdx.x * 4 + wave) * CONV_PT;
  if ((long long)blockIdx.x * 4 * CONV_PT * 16 >= R) return;         // whole workgroup beyond the minibatch
  const float* Wl = a.W + g.indW;
  for (int i = tid; i < CT * 16 * ldK; i += 256) {
    const int c = i / ldK, k = i - c * ldK;
    Ws[i] = (c < g.KnC && k < K) ? Wl[(size_t)c * K + k] : 0.f;
  }
  for (int k = tid; k < Kp; k += 256) {
    int off = 0;
    if (k < K) { const int ic = k / (g.KnY * g.KnX), f = k - ic * g.KnY * g.KnX, fy = f / g.KnX, fx = f - fy * g.KnX;
      off = ic * g.InY * g.InX + fy * g.InX + fx; }
    kOff[k] = off;
  }
  __syncthreads();
  long long rowBase[CONV_PT]; int bb[CONV_PT], pp[CONV_PT]; bool ok[CONV_PT];
#pragma unroll
  for (int t = 0; t < CONV_PT; ++t) {
    const long long r = (tile0 + t) * 16 + li;
    ok[t] = r < R;
    const long long rr = ok[t] ? r : 0;
    bb[t] = (int)(rr / P); pp[t] = (int)(rr - (long long)bb[t] * P);
    const int oy = pp[t] / g.OpX, ox = pp[t] - oy * g.OpX;
    rowBase[t] = (long long)bb[t] * g.ldIn + (long long)oy * g.S * g.InX + ox * g.S;
  }
  f32x4 acc[CT][CONV_PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < CONV_PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* in = g.in;
  constexpr int UN = 4;                       // MFMA steps whose operands are fetched together
  for (int s0 = 0; s0 < Kp / 4; s0 += UN) {
    float av[UN][CT], bv[UN][CONV_PT];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = 4 * (s0 + u) + lc;
      const bool kin = kk < Kp;
      const int ko = kin ? kOff[kk] : 0;
#pragma unroll
      for (int c = 0; c < CT; ++c) av[u][c] = kin ? Ws[(c * 16 + li) * ldK + kk] : 0.f;
#pragma unroll
      for (int t = 0; t < CONV_PT; ++t) bv[u][t] = in[rowBase[t] + ko];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < CONV_PT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], bv[u][t], acc[c][t], 0, 0, 0);
  }
  const float* Bl = a.W + g.indB;
#pragma unroll
  for (int t = 0; t < CONV_PT; ++t) {
    if (!ok[t]) continue;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = c * 16 + lc * 4 + r;
        if (ch < g.KnC) {
          const float x = acc[c][t][r] + Bl[(size_t)ch * P + pp[t]];
          const size_t o = (size_t)bb[t] * g.ldOut + (size_t)ch * P + pp[t];
          g.X[o] = x; g.Y[o] = softsignEval(x);
        }
      }
  }
}

template <int CT> static hipError_t launchConvFwdT(const ConvArgs& a, int l, int blocks, hipStream_t s) {
  const size_t lds = convFwdLds(a.L[l], CT);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_fwd_kernel<CT>, dim3(blocks), dim3(256), lds, s, a, l);
  return hipGetLastError();
}
hipError_t launch_conv_forward(const ConvArgs& a, int l, int maxRows, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const long long R = (long long)maxRows * g.P;
  const int blocks = (int)((R + 16 * 4 * CONV_PT - 1) / (16 * 4 * CONV_PT));
  const int CT = (g.KnC + 15) / 16;
  if (CT == 1) return launchConvFwdT<1>(a, l, blocks, s);
  if (CT == 2) return launchConvFwdT<2>(a, l, blocks, s);
  if (CT <= 4) return launchConvFwdT<4>(a, l, blocks, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// dX: gradient w.r.t. the input image of layer l, times act' of the layer below -> D of layer l - 1
// ---------------------------------------------------------------------------------------------------------------
template <int IT>     // 16-input-channel tiles
__global__ __launch_bounds__(256) void conv_dx_kernel(ConvArgs a, int l) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvGeo g = a.L[l];
  const ConvGeo gp = a.L[l - 1];                    // the layer whose outputs are this layer's inputs
  const int KK = g.KnC * g.KnY * g.KnX, KKp = convPad4(KK), ldKK = KKp + 4, P = g.P, Pin = g.InY * g.InX;
  float* Wx = reinterpret_cast<float*>(smem);                          // [IT*16][ldKK]   Wx[ic][(c, fy, fx)]
  int* kTab = reinterpret_cast<int*>(Wx + (size_t)IT * 16 * ldKK);     // [KKp]   c * P | fy << 20 | fx << 26   (-1: padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const long long R = (long long)a.B * Pin;
  const long long tile0 = ((long long)blockIdx.x * 4 + wave) * CONV_PT;
  const float* Wl = a.W + g.indW;
  const int fsz = g.KnY * g.KnX;
  for (int i = tid; i < IT * 16 * ldKK; i += 256) {
    const int ic = i / ldKK, kk = i - ic * ldKK;
    float w = 0.f;
    if (ic < g.InC && kk < KK) { const int c = kk / fsz, f = kk - c * fsz; w = Wl[((size_t)c * g.InC + ic) * fsz + f]; }
    Wx[i] = w;
  }
  for (int kk = tid; kk < KKp; kk += 256) {
    int v = -1;
    if (kk < KK) { const int c = kk / fsz, f = kk - c * fsz, fy = f / g.KnX, fx = f - fy * g.KnX; v = (c * P) | (fy << 20) | (fx << 26); }
    kTab[kk] = v;
  }
  __syncthreads();
  int bb[CONV_PT], qq[CONV_PT], iy[CONV_PT], ix[CONV_PT]; bool ok[CONV_PT]; long long dRow[CONV_PT];
#pragma unroll
  for (int t = 0; t < CONV_PT; ++t) {
    const long long r = (tile0 + t) * 16 + li;
    ok[t] = r < R;
    const long long rr = ok[t] ? r : 0;
    bb[t] = (int)(rr / Pin); qq[t] = (int)(rr - (long long)bb[t] * Pin);
    iy[t] = qq[t] / g.InX; ix[t] = qq[t] - iy[t] * g.InX;
    dRow[t] = (long long)bb[t] * g.ldOut;
  }
  const int S = g.S, sh = S == 1 ? 0 : (S == 2 ? 1 : (S == 4 ? 2 : 3));
  f32x4 acc[IT][CONV_PT];
#pragma unroll
  for (int c = 0; c < IT; ++c)
#pragma unroll
    for (int t = 0; t < CONV_PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* D = g.D;
  constexpr int UN = 4;
  for (int s0 = 0; s0 < KKp / 4; s0 += UN) {
    float av[UN][IT], bv[UN][CONV_PT];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = 4 * (s0 + u) + lc;
      const bool kin = kk < KKp;
      const int tab = kin ? kTab[kk] : -1;
      const int offC = tab & 0xFFFFF, fy = (tab >> 20) & 63, fx = (tab >> 26) & 31;
#pragma unroll
      for (int c = 0; c < IT; ++c) av[u][c] = kin ? Wx[(c * 16 + li) * ldKK + kk] : 0.f;
#pragma unroll
      for (int t = 0; t < CONV_PT; ++t) {
        const int oyS = iy[t] - fy, oxS = ix[t] - fx;
        const int oy = oyS >> sh, ox = oxS >> sh;
        const bool v = tab >= 0 && ok[t] && oyS >= 0 && oxS >= 0 && ((oyS | oxS) & (S - 1)) == 0 && oy < g.OpY && ox < g.OpX;
        bv[u][t] = v ? D[dRow[t] + offC + oy * g.OpX + ox] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < IT; ++c)
#pragma unroll
        for (int t = 0; t < CONV_PT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], bv[u][t], acc[c][t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < CONV_PT; ++t) {
    if (!ok[t]) continue;
#pragma unroll
    for (int c = 0; c < IT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ic = c * 16 + lc * 4 + r;
        if (ic < g.InC) {
          const size_t o = (size_t)bb[t] * gp.ldOut + (size_t)ic * Pin + qq[t];
          gp.D[o] = acc[c][t][r] * softsignDiff(gp.X[o]);
        }
      }
  }
}
__host__ __device__ inline size_t convDxLds(const ConvGeo& g, int IT) {
  const int KKp = convPad4(g.KnC * g.KnY * g.KnX);
  return (size_t)IT * 16 * (KKp + 4) * 4 + (size_t)KKp * 4;
}
template <int IT> static hipError_t launchConvDxT(const ConvArgs& a, int l, int blocks, hipStream_t s) {
  const size_t lds = convDxLds(a.L[l], IT);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dx_kernel<IT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_dx_kernel<IT>, dim3(blocks), dim3(256), lds, s, a, l);
  return hipGetLastError();
}
hipError_t launch_conv_dx(const ConvArgs& a, int l, hipStream_t s) {
  const ConvGeo& g = a.L[l];
  const long long R = (long long)a.B * g.InY * g.InX;
  const int blocks = (int)((R + 16 * 4 * CONV_PT - 1) / (16 * 4 * CONV_PT));
  const int IT = (g.InC + 15) / 16;
  if (IT == 1) return launchConvDxT<1>(a, l, blocks, s);
  if (IT == 2) return launchConvDxT<2>(a, l, blocks, s);
  if (IT <= 4) return launchConvDxT<4>(a, l, blocks, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// dW: partial filter gradients per (layer, 16 x 16 tile of [KnC][K], chunk of the batch x positions reduction)
// ---------------------------------------------------------------------------------------------------------------
constexpr int CONV_DW_MAXROWS = 2048;      // rows of one chunk (LDS tables)
__global__ __launch_bounds__(256) void conv_dw_kernel(ConvArgs a) {
  __shared__ long long sIn[CONV_DW_MAXROWS];      // offset of the patch origin of row r in the input array
  __shared__ int sD[CONV_DW_MAXROWS];             // b * ldOut + p
  __shared__ float red[4 * 256];
  int l = 0;
  for (int i = 1; i < a.nL; ++i) if ((int)blockIdx.x >= a.L[i].dwBlock0) l = i;
  const ConvGeo g = a.L[l];
  const int K = g.K, P = g.P, tilesK = (K + 15) / 16, tilesC = (g.KnC + 15) / 16;
  const int local = blockIdx.x - g.dwBlock0;
  const int chunk = local / (tilesK * tilesC), tile = local - chunk * tilesK * tilesC;
  const int ct = tile / tilesK, kt = tile - ct * tilesK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const long long R = (long long)a.B * P;
  const long long r0 = (long long)chunk * g.chunkRows;
  const int nr = (int)(R - r0 < g.chunkRows ? R - r0 : g.chunkRows);
  for (int i = tid; i < nr; i += 256) {
    const long long r = r0 + i;
    const int b = (int)(r / P), p = (int)(r - (long long)b * P), oy = p / g.OpX, ox = p - oy * g.OpX;
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
  constexpr int UN = 4;
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
hipError_t launch_conv_dw(const ConvArgs& a, int totalBlocks, hipStream_t s) {
  hipLaunchKernelGGL(conv_dw_kernel, dim3(totalBlocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// sum of the chunk partials in chunk order (+ Adam): one thread per filter weight of any layer
__global__ __launch_bounds__(256) void conv_reduce_adam_kernel(ConvArgs a, AdamHyper hyp, int fuseAdam) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  int l = 0;
  for (; l < a.nL; ++l) { const long long n = (long long)a.L[l].KnC * a.L[l].K; if (i < n) break; i -= n; }
  if (l >= a.nL) return;
  const ConvGeo& g = a.L[l];
  const size_t n = (size_t)g.KnC * g.K;
  float s = 0.f;
  for (int ch = 0; ch < g.nChunks; ++ch) s += g.part[(size_t)ch * n + i];
  a.G[g.indW + i] = s;
  if (fuseAdam) {
    AdamCoef c; c.eta = a.sc->etaEff[hyp.parity]; c.lambda = hyp.lambda; c.fac = hyp.fac;
    float w = a.Wrw[g.indW + i], m1 = a.M1[g.indW + i], m2 = a.M2[g.indW + i];
    adamStep(c, s, w, m1, m2);
    a.Wrw[g.indW + i] = w; a.M1[g.indW + i] = m1; a.M2[g.indW + i] = m2;
  }
}
hipError_t launch_conv_reduce_adam(const ConvArgs& a, const AdamHyper& hyp, int fuseAdam, hipStream_t s) {
  long long n = 0;
  for (int l = 0; l < a.nL; ++l) n += (long long)a.L[l].KnC * a.L[l].K;
  hipLaunchKernelGGL(conv_reduce_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, hyp, fuseAdam);
  return hipGetLastError();
}

}  // namespace hl
