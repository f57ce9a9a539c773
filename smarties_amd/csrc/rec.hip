// smarties_amd/csrc/rec.hip -- recurrent (LSTM) hidden layers with truncated back-propagation through time.
//
//   reference: Network/Layers/Layer_LSTM.h:78-165 (forward / backward of one step), Network/Network.h:102-193
//   (forward with the previous step as recurrent input, backProp over the time series), Approximator.h:116-173
//   (every step of the window is forwarded), ReplayMemory/MemoryBuffer.cpp:391-402 (the window: min(nnBPTTseq, t)
//   steps before the sampled one), Network/Layers/Layers.h:324-393 (parametric residual).
//
// First device version of this path: ONE workgroup per sample walks the sample's window step by step (the recurrence
// is sequential; samples are independent), gates one per thread, weights read through the L2.  It stores, per
// (sample, step) row, the operands of the weight-gradient contractions -- inputs [in | previous output] and the four
// gate deltas -- so that all weight gradients (and Adam) are formed by the same dW kernel as for dense layers, as
// X^T delta over the rows; rows of unused steps carry zero deltas.
#include "tail_dev.h"

namespace hl {

#define REC_MAXC 64       // cells per layer (4 gates x 64 = 256 threads)
#define REC_MAXIN 256     // inputs of the first layer

__device__ __forceinline__ float recSigm(float in) {     // Sigm::_eval (Functions.h:158-165), safeExp cut at 8 (Definitions.h:43)
  if (in > 0.f) return 1.f / (1.f + expf(fminf(8.f, fmaxf(-8.f, -in))));
  const float ex = expf(fminf(8.f, fmaxf(-8.f, in)));
  return ex / (1.f + ex);
}

__global__ __launch_bounds__(256) void rec_forward_kernel(RecArgs a) {
  __shared__ float sBuf[2][REC_MAXIN];                    // input of the current layer / output of the current block
  __shared__ float sPrevOut[HL_MAX_HIDDEN][REC_MAXC], sPrevSt[HL_MAX_HIDDEN][REC_MAXC];
  __shared__ float sX[4 * REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  // acting (MemoryBuffer::agentToMinibatch, MemoryBuffer.cpp:440-467): the agent's last steps, from a zero recurrent state
  const bool acting = a.actStates != nullptr;
  const int t = acting ? 0 : a.bt.t[b]; const long long slot = acting ? 0 : a.bt.slot[b];
  const int T = acting ? a.actSteps - 1 : min(a.nBPTT, t);
  const int nextRow = acting ? -1 : a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const float* W = a.W;
  for (int k = 0; k < nSteps; ++k) {
    const bool store = !acting && k <= T;
    const long long r = (long long)b * a.K + k;
    const long long sl = slot - T + k;
    if (tid < a.dS) {
      const float raw = acting ? a.actStates[(size_t)k * a.dS + tid] : a.rp.S[(size_t)sl * a.dS + tid];
      sBuf[0][tid] = (raw - a.rp.stMean[tid]) * a.rp.stScale[tid];                                   // Episode::standardizedState
    }
    __syncthreads();
    int cur = 0;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      const int nIn = L.nIn, nC = L.nC, NO = 4 * nC;
      const float* in = sBuf[cur];
      const float* Wj = W + L.indW; const float* Wr = Wj + (size_t)NO * nIn;
      if (store) {
        for (int i = tid; i < nIn; i += 256) L.A[r * L.ldA + i] = in[i];
        if (tid < nC) L.A[r * L.ldA + nIn + tid] = k > 0 ? sPrevOut[j][tid] : 0.f;
      }
      if (tid < NO) {
        float acc = W[L.indB + tid];
        for (int i = 0; i < nIn; ++i) acc += in[i] * Wj[(size_t)i * NO + tid];
        if (k > 0) for (int i = 0; i < nC; ++i) acc += sPrevOut[j][i] * Wr[(size_t)i * NO + tid];
        if (tid >= nC) acc = recSigm(acc);                 // the gates overwrite their inputs
        sX[tid] = acc;
        if (store) L.X[r * NO + tid] = acc;
      }
      __syncthreads();
      float out = 0.f, st = 0.f;
      if (tid < nC) {
        st = sX[tid] * sX[nC + tid] + (k > 0 ? sPrevSt[j][tid] * sX[2 * nC + tid] : 0.f);
        const float co = actEval(HL_FUNC_TANH, st);
        out = sX[3 * nC + tid] * co;
        if (store) { L.Y[r * NO + tid] = out; L.Y[r * NO + nC + tid] = st; L.Y[r * NO + 2 * nC + tid] = co; }
        float blk = out;                                   // ParametricResidualLayer::forward (Layers.h:347-361)
        if (L.hasRes && tid < L.resW) blk += in[tid] * W[L.indWr + tid] + W[L.indBr + tid];
        sBuf[cur ^ 1][tid] = blk;
      }
      __syncthreads();
      if (tid < nC) { sPrevOut[j][tid] = out; sPrevSt[j][tid] = st; }
      cur ^= 1;
    }
    const int nCl = a.L[a.nL - 1].nC;
    if (k == T && tid < nCl) a.Yout[(size_t)b * a.ldY + tid] = sBuf[cur][tid];
    if (k == T + 1 && tid < nCl) a.Yout[(size_t)nextRow * a.ldY + tid] = sBuf[cur][tid];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void rec_backward_kernel(RecArgs a) {
  __shared__ float sTop[2][REC_MAXIN];                    // error w.r.t. the output of the current block (from above, same step)
  __shared__ float sRec[HL_MAX_HIDDEN][REC_MAXC];          // error w.r.t. this step's LSTM output coming from step k+1
  __shared__ float sNxtSt[HL_MAX_HIDDEN][REC_MAXC], sNxtF[HL_MAX_HIDDEN][REC_MAXC];
  __shared__ float sD[4 * REC_MAXC], sRes[REC_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = a.bt.t[b];
  const int T = min(a.nBPTT, t);
  const float* W = a.W;
  // rows of the steps this sample does not have: zero deltas (their stale inputs then add nothing to the gradients)
  for (int k = T + 1; k < a.K; ++k) {
    const long long r = (long long)b * a.K + k;
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      if (tid < 4 * L.nC) L.D[r * 4 * L.nC + tid] = 0.f;
      if (L.hasRes && tid < L.nC) L.Rd[r * L.ldR + tid] = 0.f;
    }
  }
  for (int k = T; k >= 0; --k) {
    const long long r = (long long)b * a.K + k;
    int cur = 0;
    const int nCl = a.L[a.nL - 1].nC;
    if (tid < nCl) sTop[0][tid] = k == T ? a.Dres[(size_t)b * a.ldD + tid] : 0.f;
    __syncthreads();
    for (int j = a.nL - 1; j >= 0; --j) {
      const RecLayer& L = a.L[j];
      const int nIn = L.nIn, nC = L.nC, NO = 4 * nC;
      const float* Wj = W + L.indW; const float* Wr = Wj + (size_t)NO * nIn;
      if (tid < nC) {
        const float eTop = sTop[cur][tid];
        // ParametricResidualLayer::backward (Layers.h:363-393): the delta passes to the LSTM output, and through w to the block input
        if (L.hasRes) { L.Rd[r * L.ldR + tid] = eTop; sRes[tid] = tid < L.resW ? eTop * W[L.indWr + tid] : 0.f; }
        const float D = eTop + (k < T ? sRec[j][tid] : 0.f);
        // LSTMLayer::backward (Layer_LSTM.h:127-165)
        const float co = L.Y[r * NO + 2 * nC + tid];
        const float cellInpt = L.X[r * NO + tid], IG = L.X[r * NO + nC + tid], FG = L.X[r * NO + 2 * nC + tid], OG = L.X[r * NO + 3 * nC + tid];
        const float diff = (1.f - co * co) * D;
        const float sd = diff * OG + (k < T ? sNxtSt[j][tid] * sNxtF[j][tid] : 0.f);
        const float d0 = IG * sd;
        const float d1 = IG * (1.f - IG) * cellInpt * sd;
        const float d2 = k > 0 ? FG * (1.f - FG) * L.Y[(r - 1) * NO + nC + tid] * sd : 0.f;
        const float d3 = OG * (1.f - OG) * D * co;
        sD[tid] = d0; sD[nC + tid] = d1; sD[2 * nC + tid] = d2; sD[3 * nC + tid] = d3;
        L.D[r * NO + tid] = d0; L.D[r * NO + nC + tid] = d1; L.D[r * NO + 2 * nC + tid] = d2; L.D[r * NO + 3 * nC + tid] = d3;
        sNxtSt[j][tid] = sd; sNxtF[j][tid] = FG;
      }
      __syncthreads();
      // Layer::backward (Layers.h:123-188): errors to the block below (not below the first layer) and to the previous step
      if (j > 0) for (int i = tid; i < nIn; i += 256) {
        const float* row = Wj + (size_t)i * NO;
        float e = 0.f;
        for (int o = 0; o < NO; ++o) e += row[o] * sD[o];
        sTop[cur ^ 1][i] = (L.hasRes && i < L.resW ? sRes[i] : 0.f) + e;
      }
      float rec = 0.f;
      if (k > 0 && tid < nC) { const float* row = Wr + (size_t)tid * NO; for (int o = 0; o < NO; ++o) rec += row[o] * sD[o]; }
      __syncthreads();
      if (tid < nC) sRec[j][tid] = rec;
      cur ^= 1;
    }
    __syncthreads();
  }
}

hipError_t launch_rec_forward(const RecArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(rec_forward_kernel, dim3(a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_rec_backward(const RecArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(rec_backward_kernel, dim3(a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace hl
