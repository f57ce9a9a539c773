// smarties_amd/csrc/head_body.h -- the body of the head launch (head.hip: head_kernel_t) as a device function, so that the one-launch
// step of dense networks off the fused kernels (gemm16.hip: step_chain_kernel) runs the same code between its forward and backward
// chains.  Reference functions: see head.hip.
#pragma once
#include "tail_dev.h"

namespace hl {

#define HEAD_MAXOUT 136
// development time stamps of the first sample's wavefront (-DHL_HEAD_STAMPS), DevScalars::dbgT[0..]
#ifdef HL_HEAD_STAMPS
#define HSTAMP(i) do { if (row == 0 && lane == 0) const_cast<DevScalars*>(sc)->dbgT[i] = wall_clock64(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif

// HQ: hidden activations per lane.  nDense <= 8 (dimA <= 7) takes the register path for the output layer; wider action spaces
// use the generic path below.
// SPLIT = 1: one wavefront per sample, four samples per workgroup, HQ = ceil(H / 64).
// SPLIT = 4 (hidden width > 128): the four wavefronts of a workgroup share ONE sample -- each takes a quarter of the hidden
// units (HQ = ceil(H / 256)) in the output layer and in the back-propagation, partial outputs meet in LDS in wave order, the
// first wavefront does the fp64 head.  Four times the workgroups, a quarter of the serial work and of the bytes per wavefront
// (device time stamps on the RACER_atari shape, one wavefront per sample: 6.6 us until the loads are in, 2.7 us output layer,
// 3.0 us head, 4.2 us write-backs and back-propagation).
constexpr int HEAD_LDD = 136;      // staged deltas / pre-activations per sample: 1 + 2 x 64 options at most
constexpr int HEAD_LDS = 4 * HEAD_MAXOUT * 8 + 4 * HEAD_LDD * 4 + 5 * HEAD_LDD * 4;
// `row`: the minibatch row of this wavefront (SPLIT = 1) / of this workgroup (SPLIT = 4: the same for its four wavefronts; the barriers
// inside are then passed by all of them or by none)
template <int HQ, int SPLIT>
__device__ __forceinline__ void headBody(const HeadArgs& a, const int row, unsigned char* smem) {
  double (*sO)[HEAD_MAXOUT] = reinterpret_cast<double (*)[HEAD_MAXOUT]>(smem);
  float (*sDelta)[HEAD_LDD] = reinterpret_cast<float (*)[HEAD_LDD]>(smem + 4 * HEAD_MAXOUT * 8);
  float (*sXo)[HEAD_LDD] = reinterpret_cast<float (*)[HEAD_LDD]>(smem + 4 * HEAD_MAXOUT * 8 + 4 * HEAD_LDD * 4);   // pre-activations of the output layer (nnOutputFunc)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ws = SPLIT == 4 ? 0 : wave;                 // slot of this wavefront's sample in the shared arrays
  const int part = SPLIT == 4 ? wave : 0;               // which quarter of the hidden units
  const int kBase = part * 64 * HQ;
  float (*sPart)[HEAD_LDD] = reinterpret_cast<float (*)[HEAD_LDD]>(smem + 4 * HEAD_MAXOUT * 8 + 4 * HEAD_LDD * 4) + 1;   // SPLIT: partial outputs of the four wavefronts, rows 1..4 behind sXo[0]
  const DevScalars* sc = a.sc;
  HSTAMP(0);
  // (a minibatch row's replay slot is requested beside the row count, not behind the test on it: one dependent round trip less in front
  //  of the replay rows' loads below; the next-state rows' map is valid below the row count only)
  const int nRowsNow = sc->nRows[a.parity];
  long long slotEarly = 0;
  if (row < a.B) slotEarly = a.bt.slot[row];
  if (row >= nRowsNow) return;
  HSTAMP(1);
  const int B = a.B, dA = a.dA, nDense = a.nDense, H = a.H, nAdv = a.nAdv, pM = 1 + nAdv;
  const bool hasAdv = nAdv > 0 || a.nOpt > 0;
  const bool isNext = row >= B;
  const int b = isNext ? a.bt.nextSrc[row - B] : row;
  const long long slot = isNext ? a.bt.slot[b] : slotEarly;
  const float* Wo = a.params + a.indWo;
  const bool small = nDense <= 8;
  const bool mid = nDense > 8 && nDense <= 16;      // two chunks of eight outputs (RACER heads with a few options / actions): both in registers

  // ---- every load, up front -------------------------------------------------------------------
  float yv[HQ], xl[HQ], yl[HQ];
  float4 w0[HQ], w1[HQ], w2[HQ], w3[HQ];
#pragma unroll
  for (int q = 0; q < HQ; ++q) {
    const int k = kBase + lane + 64 * q;
    const bool ok = k < H;
    yv[q] = ok ? a.Yin[(size_t)row * a.ldY + k] : 0.f;
    xl[q] = (ok && !isNext) ? a.Xlast[(size_t)row * a.ldD + k] : 0.f;
    yl[q] = (ok && !isNext) ? a.Ylast[(size_t)row * a.ldD + k] : 0.f;
    w0[q] = make_float4(0.f, 0.f, 0.f, 0.f); w1[q] = w0[q]; w2[q] = w0[q]; w3[q] = w0[q];
    if (ok && (small || mid)) {
      w0[q] = *reinterpret_cast<const float4*>(Wo + (size_t)k * a.ldWo);
      w1[q] = *reinterpret_cast<const float4*>(Wo + (size_t)k * a.ldWo + 4);
    }
    if (ok && mid) {      // (a chunk fetched inside the loops below costs a dependent round trip through the L2, twice: forward and back)
      w2[q] = *reinterpret_cast<const float4*>(Wo + (size_t)k * a.ldWo + 8);
      w3[q] = *reinterpret_cast<const float4*>(Wo + (size_t)k * a.ldWo + 12);
    }
  }
  // hand the (episode, next-row) map of THIS minibatch to the bookkeeping pass, which runs while
  // the sampler already overwrites bt.eid / bt.nextOf for the next step
  // (requested here, stored with the write-backs at the end: a store right behind its load makes the wavefront wait for EVERY load
  //  issued so far -- a whole round trip in front of the replay rows' loads below)
  int eidv = 0, nxtv = 0;
  if (!isNext && lane == 0) { eidv = a.bt.eid[b]; nxtv = a.bt.nextOf[b]; }
  double beta = sc->beta; const double Cmax = sc->Cmax, Cinv = sc->Cinv;      // (no launch between the update of beta and this kernel: step_exec.h)
  const long long betaWant = a.deferBeta ? sc->nGradSteps : 0;      // (deferBeta: a rider of this launch publishes beta under this number, tail_dev.h: farBetaPhase)
  const float bo = lane < nDense ? a.params[a.indBo + lane] : 0.f;
  const float bo2 = (mid && lane < 8 && 8 + lane < nDense) ? a.params[a.indBo + 8 + lane] : 0.f;
  const float bp = lane < a.nSig ? a.params[a.indBp + lane] : 0.f;
  double act = 0, bMean = 0, bStd = 1;
  if (!isNext && a.nOpt) {     // discrete head: lane 0 holds the action message, lane j the behaviour probability of option j
    if (lane == 0) act = a.rp.A[slot];
    if (lane < a.nOpt) bMean = a.rp.MU[(size_t)slot * a.nOpt + lane];
  } else if (!isNext && lane < dA) {
    act = a.rp.A[(size_t)slot * dA + lane];
    bMean = a.rp.MU[(size_t)slot * 2 * dA + lane]; bStd = a.rp.MU[(size_t)slot * 2 * dA + dA + lane];
  }
  // lanes 0..5 fetch RET, DQ, DKL, IMPW, V, ADV of the sampled step; lanes 6,7 V, ADV of t+1 (next rows)
  float misc = 0.f;
  {
    const float* arr = nullptr; long long sl = slot;
    if (!isNext) { arr = lane == 0 ? a.rp.RET : lane == 1 ? a.rp.DQ : lane == 2 ? a.rp.DKL : lane == 3 ? a.rp.IMPW :
                         lane == 4 ? a.rp.V : lane == 5 ? a.rp.ADV : nullptr; }
    else { arr = lane == 6 ? a.rp.V : lane == 7 ? a.rp.ADV : nullptr; sl = slot + 1; }
    if (arr) misc = arr[sl];
  }

  HSTAMP(2);
  // ---- output dense layer: O[o] = b[o] + sum_k y[k] W[k][o], eight outputs at a time; the weight rows of a chunk are
  // fetched as two 16-byte loads per hidden unit, all of them in flight before the first use (the first chunk was
  // requested up front, next to the activations) ------------------------------------------------------------------
  const int nChunkOut = isNext ? 1 : (nDense + 7) / 8;
  for (int c = 0; c < nChunkOut; ++c) {
    float4 wa[HQ], wb[HQ];
    if (c == 0 && (small || mid)) {
#pragma unroll
      for (int q = 0; q < HQ; ++q) { wa[q] = w0[q]; wb[q] = w1[q]; }
    } else if (c == 1 && mid) {
#pragma unroll
      for (int q = 0; q < HQ; ++q) { wa[q] = w2[q]; wb[q] = w3[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const int k = kBase + lane + 64 * q; const bool ok = k < H;
        const float* wr = Wo + (size_t)(ok ? k : 0) * a.ldWo + 8 * c;
        wa[q] = *reinterpret_cast<const float4*>(wr); wb[q] = *reinterpret_cast<const float4*>(wr + 4);
        if (!ok) { wa[q] = make_float4(0.f, 0.f, 0.f, 0.f); wb[q] = wa[q]; }
      }
    }
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
      p[0] += yv[q] * wa[q].x; p[1] += yv[q] * wa[q].y; p[2] += yv[q] * wa[q].z; p[3] += yv[q] * wa[q].w;
      p[4] += yv[q] * wb[q].x; p[5] += yv[q] * wb[q].y; p[6] += yv[q] * wb[q].z; p[7] += yv[q] * wb[q].w;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = waveSumF(p[q]);
    float mine = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q == lane) mine = p[q];
    const int o = 8 * c + lane;
    if constexpr (SPLIT == 4) { if (lane < 8 && o < nDense) sPart[part][o] = mine; }
    else if (lane < 8 && o < nDense) {      // BaseLayer::forward of the output layer: y = f(x), f = settings nnOutputFunc (Approximator.cpp:228)
      const float x = mine + (c == 0 ? bo : ((c == 1 && mid) ? bo2 : a.params[a.indBo + o]));
      sXo[ws][o] = x; sO[ws][o] = (double)(a.outFunc == HL_FUNC_LINEAR ? x : actEval(a.outFunc, x));
    }
  }
  const bool lead = SPLIT == 1 || part == 0;              // the wavefront that does the head of this sample
  if constexpr (SPLIT == 4) {
    __syncthreads();
    if (lead) for (int o = lane; o < (isNext ? 1 : nDense); o += 64) {      // the four quarter sums in wave order, then the bias
      const float x = ((sPart[0][o] + sPart[1][o]) + (sPart[2][o] + sPart[3][o])) + (o == lane ? bo : a.params[a.indBo + o]);
      sXo[ws][o] = x; sO[ws][o] = (double)(a.outFunc == HL_FUNC_LINEAR ? x : actEval(a.outFunc, x));
    }
  }
  if (lead && lane < a.nSig) sO[ws][nDense + lane] = (double)bp;   // ParamLayer, Linear (absent for the discrete head)
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  if (isNext) {   // RACER_train.cpp:23-27: V(s_{t+1}) of a truncated episode end
    const float oV = __shfl(misc, 6, 64), oA = __shfl(misc, 7, 64);
    if (lead && lane == 0) {
      const float Vn = (float)scaleNet2V(sO[ws][0]);
      a.bt.oldNextV[b] = oV; a.bt.oldNextADV[b] = oA;
      a.rp.V[slot + 1] = Vn; a.rp.ADV[slot + 1] = 0.f; a.bt.nextV[b] = Vn;
      a.bt.O[(size_t)row * a.nOut] = sO[ws][0];
    }
    return;
  }

  HSTAMP(3);
  if (a.deferBeta) {      // the step before left its far-policy count and beta to a rider of THIS launch (POST_DEFER): wait for its number
    int spins = 0;
    while (__hip_atomic_load(&a.sc->betaSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != betaWant) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { a.sc->errFlag = 81; break; }
    }
    beta = __hip_atomic_load(&a.sc->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- head: results shared by the write-back / back-propagation tail -------------------------------------
  double xRHO = 1, xDKL = 0, xV = 0, xdQ = 0, xAval = 0; bool xfar = false; double xg0 = 0;
  if (lead) {
  if (a.nOpt) {
    // ---- discrete actions: Discrete_policy (Math/Discrete_policy.h:17-208, SoftPlus-normalised probabilities) and
    // Discrete_advantage (Math/Discrete_advantage.h:17-100); outputs [V | A x nOpt | logits x nOpt], one option per lane
    const int nOpt = a.nOpt, pA = 1, pP = 1 + nOpt;
    auto sp = [](double x) { return (x + sqrt(1 + x * x)) / 2; };
    auto spD = [](double x) { return (1 + x / sqrt(1 + x * x)) / 2; };
    const bool on = lane < nOpt;
    const int label = (int)floor(__shfl(act, 0, 64));                      // ActionInfo::actionMessage2label
    const double logit = on ? sO[ws][pP + lane] : 0.0, advJ = on ? sO[ws][pA + lane] : 0.0;
    const double unnorm = on ? sp(logit) : 0.0;
    const double norm = fmax(waveSum(unnorm), 2.220446049250313e-16);
    const double pj = unnorm / norm, mj = on ? bMean : 1.0;                 // bMean carries mu_j for this head
    const double lr = on ? log(pj / mj) : 0.0;
    const double RHO = __shfl(pj, label, 64) / __shfl(mj, label, 64);       // importanceWeight (:84-91), no clipping
    const double DKL = waveSum(on ? pj * lr : 0.0);                         // KLDivergence (:126-130)
    const float Wf = (float)RHO, Cf = (float)Cmax, iCf = (float)Cinv;
    const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);
    const double expA = waveSum(on ? pj * advJ : 0.0);
    const double Aval = __shfl(advJ, label, 64) - expA;                     // computeAdvantage (:64-70)
    const double O0 = sO[ws][0], V = scaleNet2V(O0);
    const double Qret = (double)__shfl(misc, 0, 64);
    const double A_RET = Qret - V, dQ = A_RET - Aval;
    const double g0 = far ? 0.0 : fmin(1.0, RHO) * dQ * beta * scaleVdiff(O0);
    const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);
    // KLDivGradient(mu, -1) (:152-162): sum_j tmp_j ((i == j) - p_j) = tmp_i - sum_j tmp_j p_j
    const double tmp = on ? -(1 + lr) / norm : 0.0;
    const double tp = waveSum(on ? tmp * pj : 0.0);
    if (on) {
      const double dpos = spD(logit);
      const double penal = (tmp - tp) * dpos;
      double pol = 0;
      if (!far) { const double factor = A_RET * fmin(Cmax, RHO); pol = ((lane == label ? factor / unnorm : 0.0) - factor / norm) * dpos; }   // policyGradient (:136-144)
      const double gP = beta * pol + (1 - beta) * penal;                    // penalizeReFER + makeNetworkGrad
      const double gA = Qer * ((lane == label ? 1.0 : 0.0) - pj);           // Discrete_advantage::grad (:51-58)
      sDelta[ws][pP + lane] = (float)gP; sDelta[ws][pA + lane] = (float)gA;
      a.bt.G[(size_t)b * a.nOut + pP + lane] = (double)(float)gP;
      a.bt.G[(size_t)b * a.nOut + pA + lane] = (double)(float)gA;
    }
    xRHO = RHO; xDKL = DKL; xV = V; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
  } else {
    const double MAXM = 8.31776613503286, LOG2PI_2 = 9.1893853320467266954096885456237942e-01;
    double lw = 0, kl = 0, mean = 0, stdev = 1, invStd = 1, dPos = 0;
    bool bnd = false;
    if (lane < dA) {
      const int i = lane;
      bnd = a.bounded[i] != 0;
      mean = sO[ws][pM + i];
      const double pp = sO[ws][nDense + i];
      const double rt = sqrt(1 + pp * pp);
      stdev = (pp + rt) / 2; invStd = 1 / stdev; dPos = (1 + pp / rt) / 2;
      const double bInv = 1 / bStd;
      // log pi(a) - log mu(a): the tanh Jacobian J of SquashedNormalPolicy::logProb (:240-249) and
      // the log(2 pi)/2 constants appear in both terms and cancel, log(invStd/J) - log(bInv/J) =
      // -log(stdev/bStd); the same logarithm serves the KL divergence (log CmuCpi = 2 log(stdev/bStd)).
      // One fp64 log per action component instead of three logs and a tanh (agrees with the
      // reference's term-by-term evaluation to ~1e-16 relative).
      const double m = bnd ? (mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean)) : mean;
      const double u1 = (act - m) * invStd, u2 = (act - bMean) * bInv;
      const double qq = stdev * bInv, lq = log(qq);
      lw = (u2 * u2 - u1 * u1) / 2 - lq;
      const double CmuCpi = qq * qq, dm = (mean - bMean) * bInv;
      kl = (CmuCpi - 1 + dm * dm - 2 * lq) / 2;
    }
    const double logW = waveSum(lw), DKL = waveSum(kl);
    const double RHO = exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
    const float Wf = (float)RHO, Cf = (float)Cmax, iCf = (float)Cinv;
    const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);          // Episode.h:28-33 (Fval)
    const double O0 = sO[ws][0];
    const double V = scaleNet2V(O0);
    const double Qret = (double)__shfl(misc, 0, 64);
    // Gaussian_advantage::computeAdvantage (Gaus_advantage.h:76-88): A = coef (exp(-1/2 sum (a-m)^2 / L) - ratio),
    // L = L+ above the policy mean, L- below; sums and products in the reference's component order
    double Aval = 0, advCoef = 0, advOrig = 0, advRatio = 1, p1 = 1, p2 = 1, pm = 0;
    auto sp = [](double x) { return (x + sqrt(1 + x * x)) / 2; };                 // SoftPlus::_eval (Functions.h:541-584)
    auto spD = [](double x) { return (1 + x / sqrt(1 + x * x)) / 2; };
    if (nAdv) {
      double quadI = 0, rI = 1;
      if (lane < dA) {
        p1 = sp(sO[ws][2 + lane]); p2 = sp(sO[ws][2 + dA + lane]);
        pm = bnd ? (mean > MAXM ? MAXM : (mean < -MAXM ? -MAXM : mean)) : mean;
        const double d = act - pm, S = stdev * stdev;
        quadI = d * d / (act > pm ? p1 : p2);
        rI = sqrt(p1 / (p1 + S)) / 2 + sqrt(p2 / (p2 + S)) / 2;
      }
      double quad = 0;
      for (int i = 0; i < dA; ++i) { quad += __shfl(quadI, i, 64); advRatio *= __shfl(rI, i, 64); }
      advCoef = sp(sO[ws][1]); advOrig = exp(-quad / 2);
      Aval = advCoef * (advOrig - advRatio);
    }
    const double A_RET = Qret - V, dQ = A_RET - Aval;                // Zero_advantage: A = 0
    const double Ver = fmin(1.0, RHO) * dQ;
    const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);    // RACER_train.cpp:42,56
    const double g0 = far ? 0.0 : Ver * beta * scaleVdiff(O0);
    const double coef = A_RET * fmin(Cmax, RHO);
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
      const double gM = beta * polM + (1 - beta) * penalM;
      const double gS = beta * polS + (1 - beta) * penalS;
      // Activation::addOutputDelta: nnReal += Real (Activation.h:108-117)
      sDelta[ws][pM + lane] = (float)gM;
      a.bt.gParam[(size_t)b * dA + lane] = (float)gS;
      a.bt.G[(size_t)b * a.nOut + pM + lane] = (double)(float)gM;
      a.bt.G[(size_t)b * a.nOut + nDense + lane] = (double)(float)gS;
      if (nAdv) {   // Gaussian_advantage::grad (Gaus_advantage.h:91-116) for the two precisions of this component
        const double expect = -advRatio, S = stdev * stdev, d = act - pm;
        double g1 = act > pm ? advOrig * advCoef * ((d / p1) * (d / p1)) / 2 : 0;
        double g2 = act < pm ? advOrig * advCoef * ((d / p2) * (d / p2)) / 2 : 0;
        const double F = 2 / (sqrt(p1 / (p1 + S)) + sqrt(p2 / (p2 + S)));
        const double q1 = p1 + S, q2 = p2 + S;
        g1 += F * expect * advCoef * (S / sqrt(p1 * (q1 * q1 * q1)) / 4);
        g2 += F * expect * advCoef * (S / sqrt(p2 * (q2 * q2 * q2)) / 4);
        g1 *= Qer * spD(sO[ws][2 + lane]); g2 *= Qer * spD(sO[ws][2 + dA + lane]);          // grad_matrix (:69-74)
        sDelta[ws][2 + lane] = (float)g1; sDelta[ws][2 + dA + lane] = (float)g2;
        a.bt.G[(size_t)b * a.nOut + 2 + lane] = (double)(float)g1;
        a.bt.G[(size_t)b * a.nOut + 2 + dA + lane] = (double)(float)g2;
      }
    }
    if (nAdv && lane == 0) {   // coefficient output of the Gaussian advantage
      const double gc = (advOrig - advRatio) * (Qer * spD(sO[ws][1]));
      sDelta[ws][1] = (float)gc; a.bt.G[(size_t)b * a.nOut + 1] = (double)(float)gc;
    }
    xRHO = RHO; xDKL = DKL; xV = V; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
  }
  {
    const float oDQ = __shfl(misc, 1, 64), oDKL = __shfl(misc, 2, 64), oW = __shfl(misc, 3, 64);
    const float oV = __shfl(misc, 4, 64), oADV = __shfl(misc, 5, 64);
    if (lane == 0) {
      sDelta[ws][0] = (float)xg0;
      a.bt.G[(size_t)b * a.nOut] = (double)(float)xg0;
      a.bt.rho[b] = xRHO; a.bt.dkl[b] = xDKL; a.bt.far[b] = xfar ? 1 : 0;
      a.bt.pEid[b] = eidv; a.bt.pNextOf[b] = nxtv;      // the (episode, next-row) map of THIS minibatch for the bookkeeping pass
      // write-backs (Fval casts, MiniBatch.h:161-175); old values kept for the aggregate updates
      const float E = (float)xdQ, D = (float)xDKL, Wn = (float)xRHO, Vf = (float)xV;
      a.bt.oldDQ[b] = oDQ; a.bt.oldDKL[b] = oDKL; a.bt.oldW[b] = oW; a.bt.oldV[b] = oV; a.bt.oldADV[b] = oADV;
      a.bt.newDQ[b] = E; a.bt.newDKL[b] = D; a.bt.newW[b] = Wn; a.bt.newV[b] = Vf;
      const float Qf = (float)(xAval + xV);                   // Episode::updateValues_atomic(t, V, Q): advantage = Q - V in Fval
      a.rp.DQ[slot] = E; a.rp.DKL[slot] = D; a.rp.IMPW[slot] = Wn; a.rp.V[slot] = Vf; a.rp.ADV[slot] = hasAdv ? Qf - Vf : 0.f;
      a.bt.newQ[b] = hasAdv ? Qf : Vf;
      a.bt.dq[b] = (double)E;
    }
  }
  HSTAMP(4);
  for (int o = lane; o < a.nOut; o += 64) a.bt.O[(size_t)row * a.nOut + o] = sO[ws][o];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // ---- deltas of the output layer and back-propagation into the last hidden block ----------------
  if (a.outFunc != HL_FUNC_LINEAR) {     // BaseLayer::backward: deltas *= f'(x, y) (Layer_Base.h:104-109)
    for (int o = lane; o < nDense; o += 64) sDelta[ws][o] *= actDiff(a.outFunc, sXo[ws][o], (float)sO[ws][o]);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
  for (int o = lane; o < nDense; o += 64) a.dOut[(size_t)b * a.ldDo + o] = sDelta[ws][o];
  }      // (lead)
  if constexpr (SPLIT == 4) __syncthreads();
  {
    float acc[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) acc[q] = 0.f;
    const int nCh = (nDense + 7) / 8;
    for (int c = 0; c < nCh; ++c) {
      float4 wa[HQ], wb[HQ];
      if (c == 0 && (small || mid)) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) { wa[q] = w0[q]; wb[q] = w1[q]; }
      } else if (c == 1 && mid) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) { wa[q] = w2[q]; wb[q] = w3[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
          const int k = kBase + lane + 64 * q;
          const float* wr = Wo + (size_t)(k < H ? k : 0) * a.ldWo + 8 * c;
          wa[q] = *reinterpret_cast<const float4*>(wr); wb[q] = *reinterpret_cast<const float4*>(wr + 4);
        }
      }
      float d[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) d[o] = 8 * c + o < nDense ? sDelta[ws][8 * c + o] : 0.f;
#pragma unroll
      for (int q = 0; q < HQ; ++q)
        acc[q] += ((wa[q].x * d[0] + wa[q].y * d[1]) + (wa[q].z * d[2] + wa[q].w * d[3])) +
                  ((wb[q].x * d[4] + wb[q].y * d[5]) + (wb[q].z * d[6] + wb[q].w * d[7]));
    }
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
      const int k = kBase + lane + 64 * q;
      if (k < H) {
        a.Dres[(size_t)b * a.ldD + k] = acc[q];
        a.D[(size_t)b * a.ldD + k] = acc[q] * actDiff(a.func, xl[q], yl[q]);
      }
    }
  }
  HSTAMP(5);
}

}  // namespace hl
