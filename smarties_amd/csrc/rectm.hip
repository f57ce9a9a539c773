// smarties_amd/csrc/rectm.hip -- LSTM layers wider than 64 cells, TIME-STEP-MAJOR: one launch per (layer, window step) over the whole
// minibatch, the step's gate sums as ONE product on the MFMA (round 5; VERDICT r04: "rec.hip contains no MFMA").
//
//   reference: Network/Layers/Layer_LSTM.h:77-166 (forward / backward of one step), Network/Network.h:155-193 (back-propagation over
//   the time series), Layers.h:123-188 (Layer::backward), Layers.h:347-393 (parametric residual).
//
// rec.hip's any-width kernels walk ONE sample's window per workgroup and read the layer's weights (2 MB at 2 x 256 cells) through the
// L2 for every sample and step: 2.9 ms per training step at 2 x 256 cells, batch 128, 17 steps (7 GB of L2 reads).  Here the windows
// are aligned at their first step; at global step k the rows r = b K + k of all samples that have that step form the operand of
//     gates[b][o] = bias[o] + sum_i A[r][i] W[i][o],        A[r] = [input of the step | previous output]   (the dW operand row itself)
// a [B x (nIn + nC)] x [(nIn + nC) x 4 nC] product per layer and step: the weights are read once per step, not once per sample.
//   forward   lstm_tm_prepare_kernel (window geometry per sample, the first layer's input rows), then per step and layer
//             lstm_tm_fwd_kernel: workgroup = 16 samples x 16 cells, its four wavefronts the four gates (16 x 16 x K on
//             v_mfma_f32_16x16x4_f32: A tile staged in LDS with 16-byte loads, W columns as 64-byte runs from the L2); epilogue =
//             the cell (Layer_LSTM.h:77-125), the rows kept for the backward pass and the dW launch, this step's block output into the
//             next layer's input row, this step's output into the next step's recurrent input
//   backward  (the deltas of the rows a sample does not have were zeroed by the prepare launch), then ONE launch per step (last first) and layer
//             (top first), lstm_tm_bwd_kernel: [error to the block below | error to the previous step] = D[r] W^T, 16 samples x 16
//             rows of W per workgroup, the reduction over the 4 nC deltas split over the four wavefronts (both operands as 16-byte
//             loads, the reduction index permuted inside groups of 16); its epilogue forms the cell deltas (Layer_LSTM.h:127-165)
//             whose inputs the product completes
// The windows carry one row more per sample than nnBPTTseq + 1 (RecArgs::K): a truncated episode's next state is step T + 1 of its
// window like any other (forward only; its deltas are zero).  Weight gradients: the common dW launch over all rows, as before.
#include "rec_dev.h"

namespace hl {

constexpr int TM_LDA = 4;      // padding of the staged A tile's rows (floats)

// window geometry of every sample + the first layer's input rows + zero recurrent input at the first step + zero deltas for the rows a
// sample does not have: a workgroup per (sample, window row) -- a workgroup per sample walked its eighteen rows one after the other (13 us)
__global__ __launch_bounds__(256) void lstm_tm_prepare_kernel(RecArgs a) {
  const int b = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
  const int t = a.bt.t[b]; const long long slot = a.bt.slot[b];
  const int T = min(a.nBPTT, t), nextRow = a.bt.nextOf[b];
  const int nSteps = T + 1 + (nextRow >= 0 ? 1 : 0);
  const long long r = (long long)b * a.K + k;
  // the backward launches tell the first of a tile's two producers from the second by the parity of an arrival counter: zeroed here, in
  // front of every window, so that a launch that never completed (device fault, failed replay) cannot leave a parity behind that would
  // make every later step pair values of different steps
  if (b == 0 && k == 0) for (int i = tid; i < a.tmCtrN; i += 256) a.tmCtr[i] = 0u;
  if (k == 0) {
    if (tid == 0) { a.tmT[b] = T; a.tmSteps[b] = nSteps; a.tmNext[b] = nextRow; }
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      for (int c = tid; c < L.nC; c += 256) L.A[r * L.ldA + L.nIn + c] = 0.f;
    }
  }
  if (k < nSteps) {
    const RecLayer& L0 = a.L[0];
    for (int i = tid; i < L0.nIn; i += 256) L0.A[r * L0.ldA + i] = recInputAt(a, false, b, slot, t, T, nextRow, k, i);
  }
  if (k > T) {      // rows a sample does not have (the next state's row included): zero deltas -- their stale inputs add nothing to the gradients
    for (int j = 0; j < a.nL; ++j) {
      const RecLayer& L = a.L[j];
      for (int o = tid; o < a.gates * L.nC; o += 256) L.D[r * a.gates * L.nC + o] = 0.f;
      if (L.hasRes) for (int c = tid; c < L.nC; c += 256) L.Rd[r * L.ldR + c] = 0.f;
    }
  }
}

// rollout inference of nets whose layers are wider than the per-sample kernels hold (256 cells): the agent's window as ONE sample of the
// time-step-major launches -- its geometry, the first layer's input rows from the given states, zero recurrent input at its first step
__global__ __launch_bounds__(256) void lstm_tm_prepare_act_kernel(RecArgs a) {
  const int tid = threadIdx.x, T = a.actSteps - 1;
  if (tid == 0) { a.tmT[0] = T; a.tmSteps[0] = T + 1; a.tmNext[0] = -1; }
  const RecLayer& L0 = a.L[0];
  const int dIn = L0.nIn;
  for (int e = tid; e < (T + 1) * dIn; e += 256) {
    const int k = e / dIn, i = e - k * dIn;
    L0.A[(size_t)k * L0.ldA + i] = recInputAt(a, true, 0, 0, 0, T, -1, k, i);
  }
  for (int j = 0; j < a.nL; ++j) {
    const RecLayer& L = a.L[j];
    for (int c = tid; c < L.nC; c += 256) L.A[L.nIn + c] = 0.f;
  }
}

// forward of layer j at window step k: samples with tmSteps[b] > k.  512 threads: wavefront w = gate (w & 3) x half (w >> 2) of the reduction
constexpr int TM_FNT = 512;
// One launch = one DIAGONAL of the (layer, step) grid: blockIdx.z picks (j0 + z, k0 - z) -- layer j at step k needs layer j - 1 at step k
// and its own step k - 1, both on the diagonal in front, so a window of K steps through nL layers is K + nL - 1 launches instead of K nL
// (what they write into a shared row -- the block output of the layer below, the recurrent input of the step before -- are different columns)
__global__ __launch_bounds__(TM_FNT) void lstm_tm_fwd_kernel(RecArgs a, int j0, int k0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int j = j0 + (int)blockIdx.z, k = k0 - (int)blockIdx.z;
  const RecLayer& L = a.L[j];
  if ((int)blockIdx.x * 16 >= L.nC) return;      // (the grid is as wide as the diagonal's widest layer)
  // development time stamps (-DHL_TM_STAMPS, tools/tm_stamps.py): the last layer's workgroup (0, 0) at window step 5
#ifdef HL_TM_STAMPS
#define TMSTMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && j == a.nL - 1 && k == 5) a.sc->dbgT[i] = wall_clock64(); } while (0)
#else
#define TMSTMP(i) do { } while (0)
#endif
  TMSTMP(0);
  const int nIn = L.nIn, nC = L.nC, NO = 4 * nC, Kt = nIn + nC, Kt4 = (Kt + 3) & ~3, lds = Kt4 + TM_LDA;
  float* sA = sm;                        // [16][lds]
  float* sG = sA + 16 * lds;             // [8][16][17]
  __shared__ int sAct[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  if (tid < 16) sAct[tid] = (b0 + tid < a.B && a.tmSteps[b0 + tid] > k) ? 1 : 0;
  // this wavefront's W column (gate, cell li) over its half of the rows [W_in; W_rec]: the first batch is requested in front of the A tile
  const int gate = wave & 3, half = wave >> 2;
  const int nS = Kt4 >> 2, nSh = (nS + 1) >> 1, sBeg = half * nSh, sEnd = min(nS, sBeg + nSh);
  const float* Wg = a.W + L.indW + (size_t)gate * nC + c0 + li;
  constexpr int UN = 16;
  float bv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) { const int i = min(4 * (sBeg + u) + lc, Kt - 1); bv[u] = Wg[(size_t)i * NO]; }      // (rows behind Kt: a valid row, the A element is zero)
  // what the cell's epilogue reads from memory -- the four biases, the state of the step before, the window's length: requested here, used
  // behind the products (a load issued there is a round trip of its own at the end of every launch of the chain)
  const int eRow = (tid >> 4) & 15, eC = c0 + (tid & 15), eB = min(b0 + eRow, a.B - 1);
  const float* Bi = a.W + L.indB;
  const long long eR = (long long)eB * a.K + k;
  const float bi0 = Bi[eC], bi1 = Bi[nC + eC], bi2 = Bi[2 * nC + eC], bi3 = Bi[3 * nC + eC];
  const float prevStE = k > 0 ? L.Y[(eR - 1) * NO + nC + eC] : 0.f;
  const int stepsE = a.tmSteps[eB], TE = a.tmT[eB];
  float wrE = 0.f, brE = 0.f;
  if (L.hasRes && eC < L.resW) { wrE = a.W[L.indWr + eC]; brE = a.W[L.indBr + eC]; }
  // the A tile: rows r = b K + k, Kt floats each, 16-byte pieces (the row pitch is a multiple of 16 floats); zeros behind Kt
  {
    const int q4 = Kt4 >> 2;
    for (int i = tid; i < 16 * q4; i += TM_FNT) {
      const int row = i / q4, q = i - row * q4, b = min(b0 + row, a.B - 1);
      f32x4 v = *reinterpret_cast<const f32x4*>(L.A + ((size_t)b * a.K + k) * L.ldA + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * q + e >= Kt) v[e] = 0.f;
      *reinterpret_cast<f32x4*>(sA + row * lds + 4 * q) = v;
    }
  }
  TMSTMP(1);
  __syncthreads();
  TMSTMP(2);
  bool any = false;
#pragma unroll
  for (int i = 0; i < 16; ++i) any = any || sAct[i] != 0;
  if (!any) return;
  // acc[q] = sample 4 lc + q, cell li
  const float* ar = sA + li * lds + lc;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = sBeg; s0 < sEnd; s0 += UN) {
    float bn[UN];
    const bool more = s0 + UN < sEnd;
    if (more) {
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int i = min(4 * (s0 + UN + u) + lc, Kt - 1); bn[u] = Wg[(size_t)i * NO]; }      // the next batch flies during this one's MFMAs
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float av = s0 + u < sEnd ? ar[4 * (s0 + u)] : 0.f;
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[u], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[u], acc0, 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < UN; ++u) bv[u] = bn[u];
    }
  }
  TMSTMP(3);
#pragma unroll
  for (int q = 0; q < 4; ++q) sG[(wave * 16 + 4 * lc + q) * 17 + li] = acc0[q] + acc1[q];
  __syncthreads();
  TMSTMP(4);
  // the cell of (sample row, cell c): Layer_LSTM.h:77-125
  if (tid >= 256) return;
  const int row = tid >> 4, cc = tid & 15, b = b0 + row, c = c0 + cc;
  if (!sAct[row]) return;
  const long long r = (long long)b * a.K + k;
  auto gsum = [&](int g) { return sG[(g * 16 + row) * 17 + cc] + sG[((4 + g) * 16 + row) * 17 + cc]; };
  const float ci = gsum(0) + bi0;
  const float ig = recSigm(gsum(1) + bi1);
  const float fg = recSigm(gsum(2) + bi2);
  const float og = recSigm(gsum(3) + bi3);
  const float prevSt = prevStE;
  const float st = ci * ig + prevSt * fg;
  const float co = actEval(HL_FUNC_TANH, st);
  const float out = og * co;
  L.X[r * NO + c] = ci; L.X[r * NO + nC + c] = ig; L.X[r * NO + 2 * nC + c] = fg; L.X[r * NO + 3 * nC + c] = og;
  L.Y[r * NO + c] = out; L.Y[r * NO + nC + c] = st; L.Y[r * NO + 2 * nC + c] = co;
  float blk = out;                                       // ParametricResidualLayer::forward (Layers.h:347-361)
  if (L.hasRes && c < L.resW) blk += sA[row * lds + c] * wrE + brE;
  const int steps = stepsE, T = TE;
  if (k + 1 < steps) L.A[(r + 1) * L.ldA + nIn + c] = out;          // the next step's recurrent input
  if (j + 1 < a.nL) { const RecLayer& U = a.L[j + 1]; U.A[r * U.ldA + c] = blk; }      // the layer above, same step
  else {
    if (k == T) a.Yout[(size_t)b * a.ldY + c] = blk;
    else if (k == T + 1) a.Yout[(size_t)a.tmNext[b] * a.ldY + c] = blk;
  }
  TMSTMP(5);
}

// the cell's deltas of (layer j, step k, sample b, cell c), Layer_LSTM.h:127-165: eTop = error from the block above (same step), eRec = error
// handed back by step k + 1 (zero at the sample's last step)
__device__ __forceinline__ void tmDelta(const RecArgs& a, int j, int k, int b, int c, int T, float eTop, float eRec) {
  const RecLayer& L = a.L[j];
  const int nC = L.nC, NO = 4 * nC;
  const long long r = (long long)b * a.K + k;
  if (L.hasRes) L.Rd[r * L.ldR + c] = eTop;
  const float D = eTop + (k < T ? eRec : 0.f);
  const float ci = L.X[r * NO + c], ig = L.X[r * NO + nC + c], fg = L.X[r * NO + 2 * nC + c], og = L.X[r * NO + 3 * nC + c];
  const float co = L.Y[r * NO + 2 * nC + c];
  const float prevSt = k > 0 ? L.Y[(r - 1) * NO + nC + c] : 0.f;
  const float diff = (1.f - co * co) * D;
  const float sd = diff * og + (k < T ? a.tmSD[j][(size_t)b * nC + c] * L.X[(r + 1) * NO + 2 * nC + c] : 0.f);
  L.D[r * NO + c] = ig * sd;
  L.D[r * NO + nC + c] = ig * (1.f - ig) * ci * sd;
  L.D[r * NO + 2 * nC + c] = k > 0 ? fg * (1.f - fg) * prevSt * sd : 0.f;
  L.D[r * NO + 3 * nC + c] = og * (1.f - og) * D * co;
  a.tmSD[j][(size_t)b * nC + c] = sd;
}
__device__ __forceinline__ int c0tile(int i0, int nIn, bool below) { return (below ? i0 : i0 - nIn) >> 4; }
// Layer::backward of layer j at step k (Layers.h:123-188): e[b][i] = sum_o W[i][o] D[r][o] for rows i of [W_in; W_rec], samples with T >= k - 1
// (the deltas of a step a sample does not have are zero rows: lstm_tm_prepare_kernel).  The epilogues also form the cell deltas whose inputs
// the products complete, and a launch is one ANTI-DIAGONAL of the (layer, step) grid -- blockIdx.z picks (j0 + z, k0 - z):
//   tiles i <  nIn (j > 0)   e + residual path = the error of the block below at THIS step: one of the two inputs of the deltas of (j - 1, k)
//   tiles i >= nIn           the error handed to step k - 1: the last layer forms its deltas of (j, k - 1) at once (its error from above is
//                            the head's gradient at the sample's last step, else zero); for the other layers it is the second input of the
//                            deltas of (j, k - 1), whose first one comes from the launch (j + 1, k - 1) -- a member of the SAME diagonal.
// The two producers of a (layer, 16 cells, 16 samples) tile of deltas meet at an arrival counter: the first leaves its values (agent-scope
// stores, acknowledged before the arrival), the second reads them and forms the deltas (the pattern of dw_wide_kernel's row quarters).
// Deltas formed on diagonal e are the operands of diagonal e - 1: 18 launches instead of 34 at two layers and 17 steps.
// Launch (nL - 1, nBPTT + 1) -- a step no sample has -- starts the chain with the last layer's deltas of step nBPTT.
// Both operands come straight from memory as 16-byte loads (the reduction index permuted inside groups of 16: lane group lc takes
// o = 16 G + 4 lc + e in sub-step e); wavefront w reduces over gate w's deltas, the four partial tiles meet in LDS.
__global__ __launch_bounds__(256) void lstm_tm_bwd_kernel(RecArgs a, int j0, int k0) {
  __shared__ float sR[4 * 256];
  __shared__ int sT[16];
  __shared__ unsigned sArr;
  const int j = j0 + (int)blockIdx.z, k = k0 - (int)blockIdx.z;
  const int top = a.nL - 1;
  if (k == a.nBPTT + 1 && j != top) return;      // (only the last layer has a launch at the step behind the windows)
  const RecLayer& L = a.L[j];
  const int nIn = L.nIn, nC = L.nC, NO = 4 * nC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int row0 = j > 0 ? 0 : nIn;      // (no error below the first layer)
  const int nRowW = nIn + nC;
  const int b0 = blockIdx.y * 16, i0 = row0 + blockIdx.x * 16;
  if (i0 >= nIn + (k > 0 ? nC : 0)) return;      // (the grid is as wide as the diagonal's widest member; no error to a step in front of the first)
  if (tid < 16) sT[tid] = b0 + tid < a.B ? a.tmT[b0 + tid] : -2;
  const int iw = min(i0 + li, nRowW - 1), bl = min(b0 + li, a.B - 1);
  const f32x4* wr = reinterpret_cast<const f32x4*>(a.W + L.indW + (size_t)iw * NO + wave * nC) + lc;
  const f32x4* dr = reinterpret_cast<const f32x4*>(L.D + ((size_t)bl * a.K + k) * NO + wave * nC) + lc;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nG = nC >> 4;
  constexpr int UG = 8;
  for (int g0 = 0; g0 < nG; g0 += UG) {
    f32x4 wv[UG], dv[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) { const int G = min(g0 + u, nG - 1); wv[u] = wr[4 * G]; dv[u] = dr[4 * G]; }
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      if (g0 + u < nG) {      // A = deltas (rows = samples), B = W rows (columns = rows i of W)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][0], wv[u][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][1], wv[u][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][2], wv[u][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][3], wv[u][3], acc1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sR[wave * 256 + (4 * lc + q) * 16 + li] = acc0[q] + acc1[q];
  __syncthreads();
  const int row = tid >> 4, ii = tid & 15, b = b0 + row, i = i0 + ii, T = sT[row];
  const float e = (sR[tid] + sR[256 + tid]) + (sR[512 + tid] + sR[768 + tid]);      // (zero for a sample without step k)
  // the tile's cells: (J, Kc), the same for the whole workgroup (layer inputs and cells come in multiples of 16)
  const bool below = i0 < nIn;
  const int J = below ? j - 1 : j, Kc = below ? k : k - 1, c = below ? i : i - nIn;
  const int nCJ = a.L[J].nC;
  const bool live = b < a.B && c < nCJ && T >= Kc;      // (this sample has step Kc: its rows of deltas exist)
  float v = e;
  if (below && L.hasRes && i < L.resW && b < a.B) v += L.Rd[((size_t)b * a.K + k) * L.ldR + i] * a.W[L.indWr + i];
  if (!below && j == top) {      // the last layer's deltas of the previous step: nothing else feeds them
    if (live) tmDelta(a, j, Kc, b, c, T, Kc == T ? a.Dres[(size_t)b * a.ldD + c] : 0.f, v);
    return;
  }
  // the other producer of these deltas -- (j - 1, k + 1) for a tile below, (j + 1, k - 1) else -- is a member of this diagonal, unless it
  // would lie behind the windows
  if (below && k + 1 > a.nBPTT) {
    if (live) tmDelta(a, J, Kc, b, c, T, v, 0.f);
    return;
  }
  float* mine = below ? a.tmET[J] : a.tmER[J];
  const float* other = below ? a.tmER[J] : a.tmET[J];
  const size_t at = (size_t)min(b, a.B - 1) * nCJ + min(c, nCJ - 1);
  // (hand-off without a release / acquire pair, on purpose: the ONLY data that changes hands are these values, written by agent-scope
  //  atomic stores -- which go to the coherence point themselves -- and acknowledged (vmcnt(0)) before the arrival is counted; the
  //  second producer reads them back with agent-scope atomic loads.  A release here would write back this XCD's whole L2: ~15 us.)
  if (b < a.B && c < nCJ) __hip_atomic_store(mine + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): this tile's values are at the coherence point
  __syncthreads();
  if (tid == 0) sArr = __hip_atomic_fetch_add(a.tmCtr + a.tmCtrOff[J] + (c0tile(i0, nIn, below)) * (int)gridDim.y + (int)blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if ((sArr & 1u) == 0u) return;          // the first of the two
  const float o = __hip_atomic_load(other + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (live) tmDelta(a, J, Kc, b, c, T, below ? v : o, below ? o : v);
}

// ---- MGU layers (Network/Layers/Layer_GRU.h:64-231), the same arrangement ------------------------------------------------------------
//   forget f = sigm(b_f + W_ff in + W_fr prevOut),  state s = tanh(b_s + W_sf in + W_sr (f * prevOut)),  output = f s + (1 - f) prevOut
// The state's recurrent term needs the forget gates of ALL cells: two products per (layer, step) --
//   forward   phase 0: columns [0, nC) over [in | prevOut]       -> f, and the row A2 = f * prevOut (the dW operand of W_sr)
//             phase 1: columns [nC, 2 nC) over [in | A2]          -> s, the output, the rows of the next layer / next step
//   backward  phase 0: fp = dS W_sr^T (reduction over the state deltas)        -> dF = ((s - p) dLdO + fp p) f (1 - f)
//             phase 1: [error below | g] = [dF | dS] W_in^T  |  dF W_fr^T      -> error of the block below; error to step k - 1 =
//                      (1 - f) dLdO + f fp + g.  Its epilogue forms dLdO and dS of the cell whose inputs it completes (as the LSTM
//                      kernel does): the block below at this step, the last layer at the previous step
// W is [nIn + nC][2 nC] (forget columns first); rows kept per (sample, step): X = [f | s], Y = output, D = [dF | dS].
__global__ __launch_bounds__(TM_FNT) void mgu_tm_fwd_kernel(RecArgs a, int j0, int k0, int phase) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int j = j0 + (int)blockIdx.z, k = k0 - (int)blockIdx.z;      // (a diagonal per launch: lstm_tm_fwd_kernel)
  const RecLayer& L = a.L[j];
  if ((int)blockIdx.x * 16 >= L.nC) return;
  const int nIn = L.nIn, nC = L.nC, NO = 2 * nC, Kt = nIn + nC, Kt4 = (Kt + 3) & ~3, lds = Kt4 + TM_LDA;
  float* sA = sm;                        // [16][lds]
  float* sG = sA + 16 * lds;             // [8][16][17]
  __shared__ int sAct[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  if (tid < 16) sAct[tid] = (b0 + tid < a.B && a.tmSteps[b0 + tid] > k) ? 1 : 0;
  // wavefront w: an eighth of the rows [W_in; W_rec] of column (phase, cell li)
  const int nS = Kt4 >> 2, nSe = (nS + 7) >> 3, sBeg = wave * nSe, sEnd = min(nS, sBeg + nSe);
  const float* Wg = a.W + L.indW + (size_t)phase * nC + c0 + li;
  constexpr int UN = 16;
  float bv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) { const int i = min(4 * (sBeg + u) + lc, Kt - 1); bv[u] = Wg[(size_t)i * NO]; }
  {      // A tile: [in | prevOut] (phase 0) or [in | f * prevOut] (phase 1), zeros behind Kt
    const int q4 = Kt4 >> 2;
    for (int i = tid; i < 16 * q4; i += TM_FNT) {
      const int row = i / q4, q = i - row * q4, b = min(b0 + row, a.B - 1);
      const long long r = (long long)b * a.K + k;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = 4 * q + e;
        v[e] = x < nIn ? L.A[r * L.ldA + x] : (x < Kt ? (phase ? L.A2[r * L.ldA2 + (x - nIn)] : L.A[r * L.ldA + x]) : 0.f);
      }
      *reinterpret_cast<f32x4*>(sA + row * lds + 4 * q) = v;
    }
  }
  __syncthreads();
  bool any = false;
#pragma unroll
  for (int i = 0; i < 16; ++i) any = any || sAct[i] != 0;
  if (!any) return;
  const float* ar = sA + li * lds + lc;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = sBeg; s0 < sEnd; s0 += UN) {
    if (s0 > sBeg) {
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int i = min(4 * (s0 + u) + lc, Kt - 1); bv[u] = Wg[(size_t)i * NO]; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float av = s0 + u < sEnd ? ar[4 * (s0 + u)] : 0.f;
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[u], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[u], acc0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sG[(wave * 16 + 4 * lc + q) * 17 + li] = acc0[q] + acc1[q];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, cc = tid & 15, b = b0 + row, c = c0 + cc;
  if (!sAct[row]) return;
  float sum = a.W[L.indB + phase * nC + c];
#pragma unroll
  for (int w = 0; w < 8; ++w) sum += sG[(w * 16 + row) * 17 + cc];
  const long long r = (long long)b * a.K + k;
  const float po = sA[row * lds + nIn + c];      // phase 0: prevOut; phase 1: f * prevOut
  if (phase == 0) {
    const float f = recSigm(sum);
    L.X[r * NO + c] = f;
    L.A2[r * L.ldA2 + c] = po * f;
    return;
  }
  const float st = actEval(HL_FUNC_TANH, sum);
  L.X[r * NO + nC + c] = st;
  const float f = L.X[r * NO + c], prev = L.A[r * L.ldA + nIn + c];
  const float out = k > 0 ? f * st + (1.f - f) * prev : f * st;
  L.Y[r * NO + c] = out;
  float blk = out;
  if (L.hasRes && c < L.resW) blk += sA[row * lds + c] * a.W[L.indWr + c] + a.W[L.indBr + c];
  const int steps = a.tmSteps[b], T = a.tmT[b];
  if (k + 1 < steps) L.A[(r + 1) * L.ldA + nIn + c] = out;
  if (j + 1 < a.nL) { const RecLayer& U = a.L[j + 1]; U.A[r * U.ldA + c] = blk; }
  else {
    if (k == T) a.Yout[(size_t)b * a.ldY + c] = blk;
    else if (k == T + 1) a.Yout[(size_t)a.tmNext[b] * a.ldY + c] = blk;
  }
}
// dLdO and the state delta of (layer j, step k, sample b, cell c): eTop = error from the block above, eRec = error handed back by step k + 1
__device__ __forceinline__ void tmMguOpen(const RecArgs& a, int j, int k, int b, int c, int T, float eTop, float eRec) {
  const RecLayer& L = a.L[j];
  const int nC = L.nC, NO = 2 * nC;
  const long long r = (long long)b * a.K + k;
  if (L.hasRes) L.Rd[r * L.ldR + c] = eTop;
  const float dLdO = eTop + (k < T ? eRec : 0.f);
  const float f = L.X[r * NO + c], st = L.X[r * NO + nC + c];
  a.tmSD[j][(size_t)b * nC + c] = dLdO;
  L.D[r * NO + nC + c] = dLdO * f * (1.f - st * st);
}
// (a launch is one anti-diagonal of the (layer, step) grid and one phase, blockIdx.z picks (j0 + z, k0 - z): lstm_tm_bwd_kernel)
__global__ __launch_bounds__(256) void mgu_tm_bwd_kernel(RecArgs a, int j0, int k0, int phase) {
  __shared__ float sR[4 * 256];
  __shared__ int sT[16];
  __shared__ unsigned sArr;
  const int j = j0 + (int)blockIdx.z, k = k0 - (int)blockIdx.z, top = a.nL - 1;
  if (k == a.nBPTT + 1 && (j != top || phase == 0)) return;      // (behind the windows: the last layer's phase 1 only -- it starts the chain)
  const RecLayer& L = a.L[j];
  const int nIn = L.nIn, nC = L.nC, NO = 2 * nC;
  if (phase == 0 ? (int)blockIdx.x * 16 >= nC : (j > 0 ? 0 : nIn) + (int)blockIdx.x * 16 >= nIn + (k > 0 ? nC : 0)) return;      // (the grid is as wide as the diagonal's widest member)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lc = lane >> 4;
  const int b0 = blockIdx.y * 16;
  if (tid < 16) sT[tid] = b0 + tid < a.B ? a.tmT[b0 + tid] : -2;
  const int bl = min(b0 + li, a.B - 1);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int UG = 8;
  if (phase == 0) {
    // fp[b][c] = sum_o dS[b][o] W[nIn + c][nC + o]: the four wavefronts split the nC state deltas (k == 0: no recurrent input, fp = 0)
    const int c0 = blockIdx.x * 16;
    if (k > 0) {
      const int q = nC >> 2;      // deltas per wavefront (nC is a multiple of 16)
      const f32x4* wr = reinterpret_cast<const f32x4*>(a.W + L.indW + (size_t)(nIn + c0 + li) * NO + nC + wave * q) + lc;
      const f32x4* dr = reinterpret_cast<const f32x4*>(L.D + ((size_t)bl * a.K + k) * NO + nC + wave * q) + lc;
      const int nG = q >> 4, rem = q & 15;      // whole groups of 16; nC % 64 != 0 leaves a group of `rem` (a multiple of 4)
      for (int g0 = 0; g0 < nG; g0 += UG) {
        f32x4 wv[UG], dv[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) { const int G = min(g0 + u, nG - 1); wv[u] = wr[4 * G]; dv[u] = dr[4 * G]; }
#pragma unroll
        for (int u = 0; u < UG; ++u) if (g0 + u < nG) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][0], wv[u][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][1], wv[u][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][2], wv[u][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][3], wv[u][3], acc1, 0, 0, 0);
        }
      }
      if (rem) {      // the last `rem` deltas of the wavefront's range, four per lane group: lane group lc takes element 16 nG + 4 e + lc in sub-step e < rem / 4
        const float* wq = a.W + L.indW + (size_t)(nIn + c0 + li) * NO + nC + wave * q + 16 * nG;
        const float* dq = L.D + ((size_t)bl * a.K + k) * NO + nC + wave * q + 16 * nG;
        for (int e = 0; 4 * e < rem; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[4 * e + lc], wq[4 * e + lc], acc0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sR[wave * 256 + (4 * lc + q) * 16 + li] = acc0[q] + acc1[q];
    __syncthreads();
    const int row = tid >> 4, cc = tid & 15, b = b0 + row, c = c0 + cc, T = sT[row];
    if (T < k) return;
    const float fp = k > 0 ? (sR[tid] + sR[256 + tid]) + (sR[512 + tid] + sR[768 + tid]) : 0.f;
    const long long r = (long long)b * a.K + k;
    const float f = L.X[r * NO + c], st = L.X[r * NO + nC + c], p = k > 0 ? L.Y[(r - 1) * NO + c] : 0.f;
    const float dLdO = a.tmSD[j][(size_t)b * nC + c];
    a.tmFP[j][(size_t)b * nC + c] = fp;
    L.D[r * NO + c] = ((st - p) * dLdO + fp * p) * f * (1.f - f);
    return;
  }
  // phase 1: rows i of [W_in; W_rec]: i < nIn reduce over [dF | dS] (wavefronts 0, 1 the forget half, 2, 3 the state half), i >= nIn over dF only
  const int row0 = j > 0 ? 0 : nIn, i0 = row0 + blockIdx.x * 16, nRowW = nIn + nC;
  const int iw = min(i0 + li, nRowW - 1);
  const int h = nC >> 1;                                      // deltas per wavefront: half of a gate's
  const bool stateHalf = wave >= 2;
  const f32x4* wr = reinterpret_cast<const f32x4*>(a.W + L.indW + (size_t)iw * NO + wave * h) + lc;
  const f32x4* dr = reinterpret_cast<const f32x4*>(L.D + ((size_t)bl * a.K + k) * NO + wave * h) + lc;
  const float keep = (stateHalf && i0 + li >= nIn) ? 0.f : 1.f;      // (recurrent rows take the forget deltas only)
  {
    const int nG = h >> 4, rem = h & 15;
    for (int g0 = 0; g0 < nG; g0 += UG) {
      f32x4 wv[UG], dv[UG];
#pragma unroll
      for (int u = 0; u < UG; ++u) { const int G = min(g0 + u, nG - 1); wv[u] = wr[4 * G]; dv[u] = dr[4 * G]; }
#pragma unroll
      for (int u = 0; u < UG; ++u) if (g0 + u < nG) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][0], wv[u][0] * keep, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][1], wv[u][1] * keep, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][2], wv[u][2] * keep, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][3], wv[u][3] * keep, acc1, 0, 0, 0);
      }
    }
    if (rem) {
      const float* wq = a.W + L.indW + (size_t)iw * NO + wave * h + 16 * nG;
      const float* dq = L.D + ((size_t)bl * a.K + k) * NO + wave * h + 16 * nG;
      for (int e = 0; 4 * e < rem; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[4 * e + lc], wq[4 * e + lc] * keep, acc0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sR[wave * 256 + (4 * lc + q) * 16 + li] = acc0[q] + acc1[q];
  __syncthreads();
  const int row = tid >> 4, ii = tid & 15, b = b0 + row, i = i0 + ii, T = sT[row];
  const float e = (sR[tid] + sR[256 + tid]) + (sR[512 + tid] + sR[768 + tid]);      // (zero for a sample without step k)
  // the tile's cells (J, Kc) and their two producers: as in lstm_tm_bwd_kernel
  const bool below = i0 < nIn;
  const int J = below ? j - 1 : j, Kc = below ? k : k - 1, c = below ? i : i - nIn;
  const int nCJ = a.L[J].nC;
  const bool live = b < a.B && c < nCJ && T >= Kc;
  float v = e;
  if (below) { if (L.hasRes && i < L.resW && b < a.B) v += L.Rd[((size_t)b * a.K + k) * L.ldR + i] * a.W[L.indWr + i]; }
  else {
    v = 0.f;
    if (b < a.B && c < nC && T >= k) {
      const long long r = (long long)b * a.K + k;
      const float f = L.X[r * NO + c];
      v = (1.f - f) * a.tmSD[j][(size_t)b * nC + c] + f * a.tmFP[j][(size_t)b * nC + c] + e;
    }
  }
  if (!below && j == top) {
    if (live) tmMguOpen(a, j, Kc, b, c, T, Kc == T ? a.Dres[(size_t)b * a.ldD + c] : 0.f, v);
    return;
  }
  if (below && k + 1 > a.nBPTT) {      // (no launch of the layer below behind the windows: these deltas have one producer)
    if (live) tmMguOpen(a, J, Kc, b, c, T, v, 0.f);
    return;
  }
  float* mine = below ? a.tmET[J] : a.tmER[J];
  const float* other = below ? a.tmER[J] : a.tmET[J];
  const size_t at = (size_t)min(b, a.B - 1) * nCJ + min(c, nCJ - 1);
  if (b < a.B && c < nCJ) __hip_atomic_store(mine + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) sArr = __hip_atomic_fetch_add(a.tmCtr + a.tmCtrOff[J] + c0tile(i0, nIn, below) * (int)gridDim.y + (int)blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if ((sArr & 1u) == 0u) return;          // the first of the two
  const float o = __hip_atomic_load(other + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (live) tmMguOpen(a, J, Kc, b, c, T, below ? v : o, below ? o : v);
}

static bool tmLayersOk(const RecArgs& a);
// acting (one window of a.actSteps given states) through the same launches: used where a layer is wider than the per-sample kernels hold
bool rec_tm_act_ok(const RecArgs& a) {
  if ((a.gates != 4 && a.gates != 2) || a.actStates == nullptr || a.B != 1 || a.tmSteps == nullptr || a.YoutRows != nullptr || a.Xin != nullptr || a.actSteps < 1 || a.actSteps > a.K) return false;
  bool wide = false;
  for (int j = 0; j < a.nL; ++j) wide = wide || a.L[j].nC > 256;
  return wide && tmLayersOk(a);
}
bool rec_tm_ok(const RecArgs& a) {
  if ((a.gates != 4 && a.gates != 2) || a.actStates != nullptr || a.tmSteps == nullptr || a.YoutRows != nullptr || a.DresRows != nullptr || a.K < a.nBPTT + 2) return false;
  return tmLayersOk(a);
}
static bool tmLayersOk(const RecArgs& a) {
  for (int j = 0; j < a.nL; ++j) {
    const RecLayer& L = a.L[j];
    if (L.nC % 16 || L.indW % 4 || (L.ldA & 3) || (j > 0 && L.nIn != a.L[j - 1].nC)) return false;
    if (a.gates == 2 && a.tmFP[j] == nullptr) return false;
    if ((size_t)(16 * (((L.nIn + L.nC + 3) & ~3) + TM_LDA) + 8 * 16 * 17) * 4 > 150 * 1024) return false;
  }
  return true;
}
static size_t tmFwdLds(const RecLayer& L) { return (size_t)(16 * (((L.nIn + L.nC + 3) & ~3) + TM_LDA) + 8 * 16 * 17) * 4; }
hipError_t launch_rec_tm_forward(const RecArgs& a, hipStream_t s) {
  const bool acting = a.actStates != nullptr;
  const int kLast = acting ? a.actSteps - 1 : a.nBPTT + 1;      // the last window step any sample can have
  if (acting) hipLaunchKernelGGL(lstm_tm_prepare_act_kernel, dim3(1), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(lstm_tm_prepare_kernel, dim3(a.B, a.K), dim3(256), 0, s, a);
  size_t ldsMax = 0;
  for (int j = 0; j < a.nL; ++j) ldsMax = std::max(ldsMax, tmFwdLds(a.L[j]));
  // diagonal d of the (layer, step) grid: layers jLo .. jLo + nz - 1 at steps d - j
  auto diag = [&](int d, int* jLo, int* nz, int* gx, size_t* lds) {
    *jLo = std::max(0, d - kLast); const int jHi = std::min(a.nL - 1, d);
    *nz = jHi - *jLo + 1; *gx = 0; *lds = 0;
    for (int j = *jLo; j <= jHi; ++j) { *gx = std::max(*gx, a.L[j].nC / 16); *lds = std::max(*lds, tmFwdLds(a.L[j])); }
  };
  if (a.gates == 2) {
    hipError_t e2 = ensureDynLds(reinterpret_cast<const void*>(mgu_tm_fwd_kernel), ldsMax); if (e2 != hipSuccess) return e2;
    for (int d = 0; d <= kLast + a.nL - 1; ++d) {
      int jLo, nz, gx; size_t lds; diag(d, &jLo, &nz, &gx, &lds);
      for (int ph = 0; ph < 2; ++ph)
        hipLaunchKernelGGL(mgu_tm_fwd_kernel, dim3(gx, (a.B + 15) / 16, nz), dim3(TM_FNT), lds, s, a, jLo, d - jLo, ph);
    }
    return hipGetLastError();
  }
  hipError_t e = ensureDynLds(reinterpret_cast<const void*>(lstm_tm_fwd_kernel), ldsMax); if (e != hipSuccess) return e;
  for (int d = 0; d <= kLast + a.nL - 1; ++d) {
    int jLo, nz, gx; size_t lds; diag(d, &jLo, &nz, &gx, &lds);
    hipLaunchKernelGGL(lstm_tm_fwd_kernel, dim3(gx, (a.B + 15) / 16, nz), dim3(TM_FNT), lds, s, a, jLo, d - jLo);
  }
  return hipGetLastError();
}
hipError_t launch_rec_tm_backward(const RecArgs& a, hipStream_t s) {
  if (a.gates == 2) {
    for (int e = a.nL - 1 + a.nBPTT + 1; e >= 0; --e) {      // anti-diagonals j + k = e: phase 0 of every member, then phase 1 of every member
      const int jLo = std::max(0, e - (a.nBPTT + 1)), jHi = std::min(a.nL - 1, e);
      int gx0 = 0, gx1 = 0;
      for (int j = jLo; j <= jHi; ++j) {
        const int k = e - j;
        if (k == a.nBPTT + 1 && j != a.nL - 1) continue;
        const RecLayer& L = a.L[j];
        const int row0 = j > 0 ? 0 : L.nIn, nOut = L.nIn + (k > 0 ? L.nC : 0) - row0;
        if (k != a.nBPTT + 1) gx0 = std::max(gx0, L.nC / 16);
        gx1 = std::max(gx1, (nOut + 15) / 16);
      }
      if (gx0 > 0) hipLaunchKernelGGL(mgu_tm_bwd_kernel, dim3(gx0, (a.B + 15) / 16, jHi - jLo + 1), dim3(256), 0, s, a, jLo, e - jLo, 0);
      if (gx1 > 0) hipLaunchKernelGGL(mgu_tm_bwd_kernel, dim3(gx1, (a.B + 15) / 16, jHi - jLo + 1), dim3(256), 0, s, a, jLo, e - jLo, 1);
    }
    return hipGetLastError();
  }
  for (int e = a.nL - 1 + a.nBPTT + 1; e >= 0; --e) {      // anti-diagonals j + k = e; the members: layers jLo .. jHi at steps e - j
    const int jLo = std::max(0, e - (a.nBPTT + 1)), jHi = std::min(a.nL - 1, e);
    int gx = 0;
    for (int j = jLo; j <= jHi; ++j) {
      const int k = e - j;
      if (k == a.nBPTT + 1 && j != a.nL - 1) continue;
      const RecLayer& L = a.L[j];
      const int row0 = j > 0 ? 0 : L.nIn, nOut = L.nIn + (k > 0 ? L.nC : 0) - row0;
      gx = std::max(gx, (nOut + 15) / 16);
    }
    if (gx <= 0) continue;
    hipLaunchKernelGGL(lstm_tm_bwd_kernel, dim3(gx, (a.B + 15) / 16, jHi - jLo + 1), dim3(256), 0, s, a, jLo, e - jLo);
  }
  return hipGetLastError();
}

}  // namespace hl
