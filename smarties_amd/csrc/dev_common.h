// smarties_amd/csrc/dev_common.h -- device helpers shared by the gfx950 kernels
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include "kernels.h"

namespace hl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Utilities::safeExp (Utils/FunctionUtilities.h:51-54), SMARTIES_EXP_CUT = 8 in the single-precision build (Definitions.h:43)
__device__ __forceinline__ float nnSafeExp(float v) { return expf(fminf(8.f, fmaxf(-8.f, v))); }
__device__ __forceinline__ float actEval(int f, float in) {   // Network/Layers/Functions.h
  switch (f) {
    case HL_FUNC_TANH:
      if (in > 0) { const float e = expf(-2 * in); return (1 - e) / (1 + e); }
      else        { const float e = expf( 2 * in); return (e - 1) / (1 + e); }
    case HL_FUNC_SOFTSIGN: return in / (1 + fabsf(in));
    case HL_FUNC_RELU: return in > 0 ? in : 0.f;
    case HL_FUNC_LRELU: return in > 0 ? in : 0.1f * in;                       // PRELU_FAC (Functions.h:16-18)
    case HL_FUNC_SIGM: { if (in > 0) return 1 / (1 + nnSafeExp(-in)); const float ex = nnSafeExp(in); return ex / (1 + ex); }
    case HL_FUNC_HARDSIGN: return in / sqrtf(1 + in * in);
    case HL_FUNC_SOFTPLUS: return (in + sqrtf(1 + in * in)) / 2;
    case HL_FUNC_EXPPLUS: return logf(1 + nnSafeExp(in));
    case HL_FUNC_EXP: return nnSafeExp(in);
    default: return in;
  }
}
__device__ __forceinline__ float actDiff(int f, float in, float out) {
  switch (f) {
    case HL_FUNC_TANH: return 1 - out * out;
    case HL_FUNC_SOFTSIGN: { const float d = 1 + fabsf(in); return 1 / (d * d); }
    case HL_FUNC_RELU: return in > 0 ? 1.f : 0.f;
    case HL_FUNC_LRELU: return in > 0 ? 1.f : 0.1f;
    case HL_FUNC_SIGM: return out * (1 - out);
    case HL_FUNC_HARDSIGN: { const float d = sqrtf(1 + in * in); return 1 / (d * d * d); }
    case HL_FUNC_SOFTPLUS: return (1 + in / sqrtf(1 + in * in)) / 2;
    case HL_FUNC_EXPPLUS: return 1 / (1 + nnSafeExp(-in));
    case HL_FUNC_EXP: return out;
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

__device__ __forceinline__ double scaleNet2V(double x) {   // Learners/RACER_common.cpp:23-27
  return x > 0 ? 100 * (x + 51) - 100 * sqrt(2601 + 100 * x) : 100 * (x - 51) + 100 * sqrt(2601 - 100 * x);
}
__device__ __forceinline__ double scaleVdiff(double x) {   // Learners/RACER_common.cpp:28-32
  return x > 0 ? 100 - 5000 / sqrt(2601 + 100 * x) : 100 - 5000 / sqrt(2601 - 100 * x);
}

// Adam::step (Network/Optimizer.cpp:61-108) with SMARTIES_NESTEROV_ADAM, SMARTIES_SAFE_ADAM,
// SMARTIES_ADAMW (Settings/Bund.h); eta already carries the bias correction (Optimizer.cpp:66)
struct AdamCoef { float eta, lambda, fac; };
// step size of the Adam step that follows `nStepDone` completed ones (Optimizer.cpp:66,132-135):
// annealed learn rate times sqrt(1-beta2^t)/(1-beta1^t), all in nnReal as the reference does
__device__ __forceinline__ float adamEtaEff(long long nStepDone, double bt1, double bt2, float eta0, double epsAnneal) {
  const long long nStep = nStepDone + 1;    // prepare_update incremented it before apply_update
  const float _eta = (float)((double)eta0 / (1 + (double)(float)nStep * epsAnneal));
  const float b1 = (float)bt1, b2 = (float)bt2;
  return _eta * sqrtf(1 - b2) / (1 - b1);
}
__device__ __forceinline__ void adamStep(const AdamCoef& c, float g, float& w, float& m1, float& m2) {
#pragma clang fp contract(off)   // same rounding wherever this is inlined (dW epilogue, stand-alone Adam kernel)
  const float B1 = 0.9f, B2 = 0.999f;
  const float penal = -w * c.lambda;
  const float DW = c.fac * g;
  m1 = B1 * m1 + (1 - B1) * DW;
  m2 = B2 * m2 + (1 - B2) * DW * DW;
  const float numer = B1 * m1 + (1 - B1) * DW;
  m2 = m2 < m1 * m1 ? m1 * m1 : m2;
  const float ret = numer / (FLT_EPSILON + sqrtf(m2));
  w = w + c.eta * (ret + penal);
}

}  // namespace hl
