// smarties_amd/csrc/rec_dev.h -- device helpers shared by the recurrent kernels (rec.hip: a workgroup or wavefront per sample;
// rectm.hip: a launch per (layer, window step) over the whole minibatch)
#pragma once
#include "dev_common.h"

namespace hl {

__device__ __forceinline__ float recSigm(float in) {     // Sigm::_eval (Functions.h:158-165), safeExp cut at 8 (Definitions.h:43)
  // (one exponential for both branches of the reference: the argument is -|in| cut at -8 either way)
  const float ex = expf(fmaxf(-8.f, -fabsf(in)));
  return in > 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
}

// Element e of the network input at step k of sample b's window (general form; T = steps in front of the sampled one, t its index
// in the episode):
//   Xin != nullptr   rows written by launches in front of this one (a convolutional stack): row b K + k, the next state's row behind
//                    the B K window rows (row B K + nextRow - B)
//   acting           the agent's last states, oldest first, `actCtx` of them in front of the window (they only feed appended
//                    observations); steps before the first given one repeat it
//   otherwise        Episode::standardizedState (Episode.h:172-183): the state of the step followed by the nApp ones before it,
//                    steps before the episode's first repeat the first
__device__ __forceinline__ float recInputAt(const RecArgs& a, bool acting, int b, long long slot, int t, int T, int nextRow, int k, int e) {
  if (a.Xin) { const long long row = k <= T ? (long long)b * a.K + k : (long long)a.B * a.K + (nextRow - a.B); return a.Xin[row * a.ldXin + e]; }
  const int j = e / a.dS, i = e - j * a.dS;
  float raw;
  if (acting) { const int g = a.actCtx + k - j; raw = a.actStates[(size_t)(g > 0 ? g : 0) * a.dS + i]; }
  else { const int tt = t - T + k, back = j < tt ? j : tt; raw = a.rp.S[(size_t)(slot - T + k - back) * a.dS + i]; }
  return (raw - a.rp.stMean[i]) * a.rp.stScale[i];
}


}  // namespace hl
