// smarties_amd/csrc/head_rows.h -- the RACER / V-RACER head with one (sample, action component or option) per lane of a 16-lane
// row, and the output-layer contraction in front of it: shared by the panel kernel (mlp_panel.hip) and the wide fused kernel
// (fusedw.hip).  The arithmetic is head.hip's (RACER::Train, Learners/RACER_train.cpp:14-67; Continuous_policy,
// Math/Continuous_policy.h:68-378, 569-810; Gaussian_advantage, Math/Gaus_advantage.h:17-127; Discrete_policy / Discrete_advantage,
// Math/Discrete_policy.h:19-208, Discrete_advantage.h:17-96; MiniBatch::setMseDklImpw / setValues, MiniBatch.h:161-175).
#pragma once
#include "tail_dev.h"

namespace hl {

__device__ __forceinline__ double spD64(double x) { return (x + sqrt(1 + x * x)) / 2; }            // SoftPlus::_eval (Functions.h:541-584)
__device__ __forceinline__ double spDiff64(double x) { return (1 + x / sqrt(1 + x * x)) / 2; }

// output layer of the panel: wave `wave` takes hidden units [wave KW, (wave + 1) KW), NT column tiles of 16 outputs; partial
// tiles -> red[(wave NT + t)][16 x 16].  pA: this lane's row of the Y panel at its first k; pB: W_out row of that k at output
// column min(li, ldWo - 1) -- columns beyond ldWo feed result columns nobody reads
template <int NT>
__device__ __forceinline__ void panelOutMma(const float* pA, const float* sWoK, int ldWo, int li, int steps, float* redW) {
  f32x4 acc[NT];
  const float* pB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; const int o = t * 16 + li; pB[t] = sWoK + (o < ldWo ? o : ldWo - 1); }
  const int stride = 4 * ldWo;
  constexpr int UN = NT <= 2 ? 8 : 4;             // steps whose operands are in flight together (one exposed LDS latency per batch)
  for (int s0 = 0; s0 < steps; s0 += UN) {
    float av[UN], bv[NT][UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int sc_ = s0 + u < steps ? s0 + u : steps - 1;      // (clamped: no predicated loads; the surplus steps multiply by zero)
      av[u] = pA[4 * sc_];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t][u] = pB[t][(size_t)sc_ * stride];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float a_ = s0 + u < steps ? av[u] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, bv[t][u], acc[t], 0, 0, 0);
    }
  }
  const int lane = threadIdx.x & 63, lc = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) redW[t * 256 + (lc * 4 + r) * 16 + li] = acc[t][r];
  }
}


// per-lane state of (sample row, components en + 16 j): the replay row, the terms that do not depend on this step's outputs, the head
template <int NCH> struct HeadRow {
  double act[NCH], bMean[NCH], bStd[NCH];
  double stdev[NCH], invStd[NCH], dPos[NCH], bInv[NCH], invVarMu[NCH], u2[NCH], lq[NCH], CmuCpi[NCH]; bool bnd[NCH], onC[NCH];
  double actMsg; float misc;

  // the sample's replay rows (depend on the slot only: requested as early as the hosting kernel can)
  __device__ __forceinline__ void load(const HeadArgs& a, bool rowValid, bool isNext, long long slot, int en) {
    const bool live = rowValid && !isNext; const int dA = a.dA, nOpt = a.nOpt;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = en + 16 * j;
      act[j] = 0; bMean[j] = nOpt ? 1.0 : 0.0; bStd[j] = 1;
      if (live) {
        if (nOpt) { if (c < nOpt) bMean[j] = a.rp.MU[(size_t)slot * nOpt + c]; }       // behaviour probability of option c
        else if (c < dA) { act[j] = a.rp.A[(size_t)slot * dA + c]; bMean[j] = a.rp.MU[(size_t)slot * 2 * dA + c]; bStd[j] = a.rp.MU[(size_t)slot * 2 * dA + dA + c]; }
      }
    }
    actMsg = 0;
    if (live && nOpt && en == 0) actMsg = a.rp.A[slot];                                 // discrete head: the action message (label + 0.1)
    misc = 0.f;
    if (rowValid) {   // lanes 0..5: RET, DQ, DKL, IMPW, V, ADV of the sampled step; next rows: lanes 6, 7: V, ADV of t+1
      const float* arr = nullptr; long long sl = slot;
      if (!isNext) arr = en == 0 ? a.rp.RET : en == 1 ? a.rp.DQ : en == 2 ? a.rp.DKL : en == 3 ? a.rp.IMPW : en == 4 ? a.rp.V : en == 5 ? a.rp.ADV : nullptr;
      else { arr = en == 6 ? a.rp.V : en == 7 ? a.rp.ADV : nullptr; sl = slot + 1; }
      if (arr) misc = arr[sl];
    }
  }
  // head terms that do not depend on this step's network outputs (the policy's standard deviation comes from the ParamLayer bias
  // alone; behaviour-policy terms from the replay rows)
  __device__ __forceinline__ void hoist(const HeadArgs& a, unsigned long long boundedMask, const float (&bpv)[NCH], bool live, int en) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = en + 16 * j; onC[j] = live && !a.nOpt && c < a.dA;
      stdev[j] = 1; invStd[j] = 1; dPos[j] = 0; bInv[j] = 1; invVarMu[j] = 1; u2[j] = 0; lq[j] = 0; CmuCpi[j] = 1; bnd[j] = false;
      if (onC[j]) {
        bnd[j] = ((boundedMask >> c) & 1ull) != 0;
        const double pp = (double)bpv[j];
        const double rt = sqrt(1 + pp * pp);
        stdev[j] = (pp + rt) / 2; invStd[j] = 1 / stdev[j]; dPos[j] = (1 + pp / rt) / 2;
        bInv[j] = 1 / bStd[j]; invVarMu[j] = 1 / (bStd[j] * bStd[j]);
        u2[j] = (act[j] - bMean[j]) * bInv[j];
        const double qq = stdev[j] * bInv[j];
        lq[j] = log(qq); CmuCpi[j] = qq * qq;
      }
    }
  }
  // the head proper.  O: this sample's outputs [nOut] (fp64, LDS); dRow: its output-layer deltas (LDS, zero on entry); xoRow: the
  // pre-activations of the output layer; mRow: the eight per-step values fetched by lanes 0..7 (misc); tq / tr: 64 doubles each of
  // scratch per sample (Gaussian advantage).  `writer`: this workgroup publishes the sample's results.
  __device__ __forceinline__ void compute(const HeadArgs& a, const double* O, float* dRow, const float* xoRow, const float* mRow, double* tq, double* tr,
                                          bool rowValid, bool isNext, bool writer, int b, long long slot, int row, int en,
                                          double beta, double Cmax, double Cinv, double actLabelMsg /* the action message of the sample (lane 0's actMsg) */) {
    const bool live = rowValid && !isNext;
    const int dA = a.dA, nDense = a.nDense, nAdv = a.nAdv, nOpt = a.nOpt, pM = 1 + nAdv, nOut = a.nOut;
    const bool hasAdv = nAdv > 0 || nOpt > 0;
    const double MAXM = 8.31776613503286;
    const double O0 = O[0];
    if (rowValid && isNext) {     // RACER_train.cpp:23-27: V(s_{t+1}) of a truncated episode end
      if (writer && en == 0) {
        const float Vn = (float)scaleNet2V(O0);
        a.bt.oldNextV[b] = mRow[6]; a.bt.oldNextADV[b] = mRow[7];
        a.rp.V[slot + 1] = Vn; a.rp.ADV[slot + 1] = 0.f; a.bt.nextV[b] = Vn;
        a.bt.O[(size_t)row * nOut] = O0;
      }
    }
    {
      const double V = scaleNet2V(O0);
      const double Qret = (double)mRow[0];
      const float Cf = (float)Cmax, iCf = (float)Cinv;
      double xRHO = 1, xDKL = 0, xdQ = 0, xAval = 0, xg0 = 0; bool xfar = false;
      if (nOpt) {
        // ---- discrete actions: Discrete_policy (SoftPlus-normalised probabilities) and Discrete_advantage; outputs
        // [V | A x nOpt | logits x nOpt], option en + 16 j per lane ----
        const int pA = 1, pP = 1 + nOpt;
        const int label = (int)floor(actLabelMsg);                                   // ActionInfo::actionMessage2label
        double logit[NCH], advJ[NCH], unnorm[NCH]; bool on[NCH];
        double su = 0;
  #pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = en + 16 * j; on[j] = live && c < nOpt;
          logit[j] = on[j] ? O[pP + c] : 0.0; advJ[j] = on[j] ? O[pA + c] : 0.0;
          unnorm[j] = on[j] ? spD64(logit[j]) : 0.0; su += unnorm[j];
        }
        const double norm = fmax(sum16(su), 2.220446049250313e-16);
        double pj[NCH], lr[NCH], sKl = 0, sEa = 0, sPl = 0, sMl = 0, sAl = 0, sTp = 0, tmp[NCH];
  #pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = en + 16 * j;
          pj[j] = unnorm[j] / norm;
          const double mj = on[j] ? bMean[j] : 1.0;
          lr[j] = on[j] ? log(pj[j] / mj) : 0.0;
          sKl += on[j] ? pj[j] * lr[j] : 0.0; sEa += on[j] ? pj[j] * advJ[j] : 0.0;
          const bool isL = on[j] && c == label;
          sPl += isL ? pj[j] : 0.0; sMl += isL ? mj : 0.0; sAl += isL ? advJ[j] : 0.0;
          tmp[j] = on[j] ? -(1 + lr[j]) / norm : 0.0; sTp += on[j] ? tmp[j] * pj[j] : 0.0;
        }
        const double RHO = sum16(sPl) / sum16(sMl);                                // importanceWeight (Discrete_policy.h:84-91), no clipping
        const double DKL = sum16(sKl);                                             // KLDivergence (:126-130)
        const float Wf = (float)RHO;
        const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);
        const double Aval = sum16(sAl) - sum16(sEa);                               // computeAdvantage (Discrete_advantage.h:64-70)
        const double tp = sum16(sTp);
        const double A_RET = Qret - V, dQ = A_RET - Aval;
        const double g0 = far ? 0.0 : fmin(1.0, RHO) * dQ * beta * scaleVdiff(O0);
        const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);
  #pragma unroll
        for (int j = 0; j < NCH; ++j) if (on[j]) {
          const int c = en + 16 * j;
          const double dpos = spDiff64(logit[j]);
          const double penal = (tmp[j] - tp) * dpos;                               // KLDivGradient(mu, -1) (:152-162)
          double pol = 0;
          if (!far) { const double factor = A_RET * fmin(Cmax, RHO); pol = ((c == label ? factor / unnorm[j] : 0.0) - factor / norm) * dpos; }   // policyGradient (:136-144)
          const double gP = beta * pol + (1 - beta) * penal;                       // penalizeReFER + makeNetworkGrad
          const double gA = Qer * ((c == label ? 1.0 : 0.0) - pj[j]);              // Discrete_advantage::grad (:51-58)
          dRow[pP + c] = (float)gP; dRow[pA + c] = (float)gA;
          if (writer) { a.bt.G[(size_t)b * nOut + pP + c] = (double)(float)gP; a.bt.G[(size_t)b * nOut + pA + c] = (double)(float)gA; }
        }
        xRHO = RHO; xDKL = DKL; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
      } else {
        double mean[NCH], pm[NCH]; bool on[NCH];
        double sLw = 0, sKl = 0;
  #pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = en + 16 * j; on[j] = onC[j];
          mean[j] = 0; pm[j] = 0;
          if (on[j]) {
            mean[j] = O[pM + c];
            // log pi(a) - log mu(a) and D_KL(pi || mu) share one logarithm (see head.hip)
            pm[j] = bnd[j] ? (mean[j] > MAXM ? MAXM : (mean[j] < -MAXM ? -MAXM : mean[j])) : mean[j];
            const double u1 = (act[j] - pm[j]) * invStd[j];
            sLw += (u2[j] * u2[j] - u1 * u1) / 2 - lq[j];
            const double dm = (mean[j] - bMean[j]) * bInv[j];
            sKl += (CmuCpi[j] - 1 + dm * dm - 2 * lq[j]) / 2;
          }
        }
        const double logW = sum16(sLw), DKL = sum16(sKl);
        const double RHO = exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
        const float Wf = (float)RHO;
        const bool far = (Cf > 1.f) && (Wf > Cf || Wf < iCf);          // Episode.h:28-33 (Fval)
        // Gaussian_advantage::computeAdvantage (Gaus_advantage.h:76-88): A = coef (exp(-1/2 sum (a-m)^2 / L) - ratio), sums and
        // products in the reference's component order (through LDS: every lane of the row walks the components)
        double Aval = 0, advCoef = 0, advOrig = 0, advRatio = 1, p1[NCH], p2[NCH];
        if (nAdv) {
  #pragma unroll
          for (int j = 0; j < NCH; ++j) {
            const int c = en + 16 * j; p1[j] = 1; p2[j] = 1;
            if (on[j]) {
              p1[j] = spD64(O[2 + c]); p2[j] = spD64(O[2 + dA + c]);
              const double d = act[j] - pm[j], S = stdev[j] * stdev[j];
              tq[c] = d * d / (act[j] > pm[j] ? p1[j] : p2[j]);
              tr[c] = sqrt(p1[j] / (p1[j] + S)) / 2 + sqrt(p2[j] / (p2[j] + S)) / 2;
            }
          }
          __builtin_amdgcn_wave_barrier(); __threadfence_block(); __builtin_amdgcn_wave_barrier();      // (a sample's 16 lanes share a wavefront)
          double quad = 0;
          if (live) for (int i = 0; i < dA; ++i) { quad += tq[i]; advRatio *= tr[i]; }
          advCoef = spD64(O[1]); advOrig = exp(-quad / 2);
          Aval = advCoef * (advOrig - advRatio);
        }
        const double A_RET = Qret - V, dQ = A_RET - Aval;                // Zero_advantage: A = 0
        const double Ver = fmin(1.0, RHO) * dQ;
        const double Qer = far ? 0.0 : beta * (fmin(Cmax, RHO) * dQ);    // RACER_train.cpp:42,56
        const double g0 = far ? 0.0 : Ver * beta * scaleVdiff(O0);
        const double coef = A_RET * fmin(Cmax, RHO);
  #pragma unroll
        for (int j = 0; j < NCH; ++j) if (on[j]) {
          const int c = en + 16 * j;
          const double penalM = -1 * ((mean[j] - bMean[j]) * invVarMu[j]);
          const double penalS = dPos[j] * -1 * ((invVarMu[j] - invStd[j] * invStd[j]) * stdev[j]);
          double polM = 0, polS = 0;
          if (!far) {
            if (bnd[j]) {
              const double dLogPdMean = (act[j] - mean[j]) * invStd[j] * invStd[j];
              const double u = (act[j] - pm[j]) * invStd[j];
              polS = dPos[j] * coef * ((u * u - 1) * invStd[j]);
              if (mean[j] >= MAXM && coef * dLogPdMean > 0) polM = 0;
              else if (mean[j] <= -MAXM && coef * dLogPdMean < 0) polM = 0;
              else polM = coef * dLogPdMean;
            } else {
              const double u = (act[j] - mean[j]) * invStd[j];
              polM = coef * (u * invStd[j]);
              polS = dPos[j] * coef * ((u * u - 1) * invStd[j]);
            }
          }
          const double gM = beta * polM + (1 - beta) * penalM;
          const double gS = beta * polS + (1 - beta) * penalS;
          dRow[pM + c] = (float)gM;                              // Activation::addOutputDelta: nnReal += Real (Activation.h:108-117)
          if (writer) {
            a.bt.gParam[(size_t)b * dA + c] = (float)gS;
            a.bt.G[(size_t)b * nOut + pM + c] = (double)(float)gM;
            a.bt.G[(size_t)b * nOut + nDense + c] = (double)(float)gS;
          }
          if (nAdv) {   // Gaussian_advantage::grad (Gaus_advantage.h:91-116) for the two precisions of this component
            const double expect = -advRatio, S = stdev[j] * stdev[j], d = act[j] - pm[j];
            double g1 = act[j] > pm[j] ? advOrig * advCoef * ((d / p1[j]) * (d / p1[j])) / 2 : 0;
            double g2 = act[j] < pm[j] ? advOrig * advCoef * ((d / p2[j]) * (d / p2[j])) / 2 : 0;
            const double F = 2 / (sqrt(p1[j] / (p1[j] + S)) + sqrt(p2[j] / (p2[j] + S)));
            const double q1 = p1[j] + S, q2 = p2[j] + S;
            g1 += F * expect * advCoef * (S / sqrt(p1[j] * (q1 * q1 * q1)) / 4);
            g2 += F * expect * advCoef * (S / sqrt(p2[j] * (q2 * q2 * q2)) / 4);
            g1 *= Qer * spDiff64(O[2 + c]); g2 *= Qer * spDiff64(O[2 + dA + c]);          // grad_matrix (:69-74)
            dRow[2 + c] = (float)g1; dRow[2 + dA + c] = (float)g2;
            if (writer) { a.bt.G[(size_t)b * nOut + 2 + c] = (double)(float)g1; a.bt.G[(size_t)b * nOut + 2 + dA + c] = (double)(float)g2; }
          }
        }
        if (nAdv && live && en == 0) {   // coefficient output of the Gaussian advantage
          const double gc = (advOrig - advRatio) * (Qer * spDiff64(O[1]));
          dRow[1] = (float)gc;
          if (writer) a.bt.G[(size_t)b * nOut + 1] = (double)(float)gc;
        }
        xRHO = RHO; xDKL = DKL; xdQ = dQ; xAval = Aval; xfar = far; xg0 = g0;
      }
      if (live && en == 0) {
        dRow[0] = (float)xg0;
        if (writer) {
          a.bt.pEid[b] = a.bt.eid[b]; a.bt.pNextOf[b] = a.bt.nextOf[b];      // (the sampler of the next step overwrites eid / nextOf meanwhile)
          a.bt.G[(size_t)b * nOut] = (double)(float)xg0;
          a.bt.rho[b] = xRHO; a.bt.dkl[b] = xDKL; a.bt.far[b] = xfar ? 1 : 0;
          // write-backs (Fval casts, MiniBatch.h:161-175); old values kept for the aggregate updates
          const float E = (float)xdQ, D = (float)xDKL, Wn = (float)xRHO, Vf = (float)V;
          a.bt.oldDQ[b] = mRow[1]; a.bt.oldDKL[b] = mRow[2]; a.bt.oldW[b] = mRow[3]; a.bt.oldV[b] = mRow[4]; a.bt.oldADV[b] = mRow[5];
          a.bt.newDQ[b] = E; a.bt.newDKL[b] = D; a.bt.newW[b] = Wn; a.bt.newV[b] = Vf;
          const float Qf = (float)(xAval + V);                    // Episode::updateValues_atomic(t, V, Q): advantage = Q - V in Fval
          a.rp.DQ[slot] = E; a.rp.DKL[slot] = D; a.rp.IMPW[slot] = Wn; a.rp.V[slot] = Vf; a.rp.ADV[slot] = hasAdv ? Qf - Vf : 0.f;
          a.bt.newQ[b] = hasAdv ? Qf : Vf;
          a.bt.dq[b] = (double)E;
        }
      }
      if (live && writer) for (int o = en; o < nOut; o += 16) a.bt.O[(size_t)row * nOut + o] = O[o];
    }
    __builtin_amdgcn_wave_barrier(); __threadfence_block(); __builtin_amdgcn_wave_barrier();
    // deltas of the output layer: BaseLayer::backward, deltas *= f'(x, y) (Layer_Base.h:104-109)
    if (live) {
      for (int o = en; o < nDense; o += 16) {
        float d = dRow[o];
        if (a.outFunc != HL_FUNC_LINEAR) { d *= actDiff(a.outFunc, xoRow[o], (float)O[o]); dRow[o] = d; }
        if (writer) a.dOut[(size_t)b * a.ldDo + o] = d;
      }
    }
  }
};

}  // namespace hl
