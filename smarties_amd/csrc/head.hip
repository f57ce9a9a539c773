// smarties_amd/csrc/head.hip -- output layer + V-RACER / ReF-ER head, one wavefront per sample.
//
//   output InnerProduct layer (Linear, Layer_Base.h:64-95) + ParamLayer (Layers.h:510-520),
//   RACER::Train (Learners/RACER_train.cpp:14-67) in fp64 with Continuous_policy
//   (Math/Continuous_policy.h:68-378, 569-738) and Zero_advantage (VRACER) or Gaussian_advantage
//   (Math/Gaus_advantage.h:17-127; outputs [V | coef, L+ x dA, L- x dA | mean x dA | sigma parameter x dA]), write-backs MiniBatch::setMseDklImpw / setValues
//   (MiniBatch.h:161-175) and the backward of the output layer into the last hidden block
//   (Layers.h:123-160).
//
// The kernel is latency bound (a few KB per sample, ~1 kFLOP of fp64 transcendental math), so
// it is written to expose ONE round of global-memory latency: every load a sample needs -- its
// hidden activations, the whole output-layer weight matrix slice of the lane, action, behaviour
// policy, Retrace target and the per-step values that are about to be overwritten -- is issued
// up front; the weight slice stays in registers and is reused for the back-propagation.
#include "head_body.h"

namespace hl {

// the kernel's body: head_body.h (shared with gemm16.hip: step_chain_kernel)
template <int HQ, int SPLIT>
__global__ __launch_bounds__(256) void head_kernel_t(HeadArgs a, ExtraArgs extra) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[HEAD_LDS > TAIL_LDS_BYTES ? HEAD_LDS : TAIL_LDS_BYTES];
  // horizontal fusion: workgroup 0 (dispatched first) runs sampler phase C of the next step
  // ... and, for wide states, workgroups 1..helpers gather the minibatch it found (one workgroup keeps only a few dozen HBM
  // misses in flight: 2 x 256 rows of 257 floats took 60 us that way)
  const int nExtra = extra.role ? 1 + extra.helpers : 0;
  if (extra.role && blockIdx.x == 0) { runExtra(extra, smem); return; }
  if ((int)blockIdx.x < nExtra) { gatherHelper(extra.samp, blockIdx.x - 1, extra.helpers, smem); return; }
  const int row = SPLIT == 4 ? (int)(blockIdx.x - nExtra) : (int)(blockIdx.x - nExtra) * 4 + (int)(threadIdx.x >> 6);
  headBody<HQ, SPLIT>(a, row, smem);
}

hipError_t launch_head(const HeadArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s) {
  ExtraArgs ex{}; if (extra) ex = *extra;
  const int nEx = ex.role ? 1 + ex.helpers : 0;
  const dim3 block(256);
  if (a.H > 128 && a.H <= 512 && maxRows >= 4096 && !nEx) {       // large batches: throughput, not latency -- a wavefront per sample, four samples per workgroup
    const dim3 grid((maxRows + 3) / 4);
    const int HQ = (a.H + 63) / 64;
    if (HQ <= 4) hipLaunchKernelGGL((head_kernel_t<4, 1>), grid, block, 0, s, a, ex);
    else hipLaunchKernelGGL((head_kernel_t<8, 1>), grid, block, 0, s, a, ex);
    return hipGetLastError();
  }
  if (a.H > 128) {       // one sample per workgroup, a quarter of the hidden units per wavefront
    const dim3 grid(maxRows + nEx);
    const int HQ = (a.H + 255) / 256;
    if (HQ <= 1) hipLaunchKernelGGL((head_kernel_t<1, 4>), grid, block, 0, s, a, ex);
    else if (HQ <= 2) hipLaunchKernelGGL((head_kernel_t<2, 4>), grid, block, 0, s, a, ex);
    else if (HQ <= 4) hipLaunchKernelGGL((head_kernel_t<4, 4>), grid, block, 0, s, a, ex);
    else if (HQ <= 8) hipLaunchKernelGGL((head_kernel_t<8, 4>), grid, block, 0, s, a, ex);
    else return hipErrorInvalidValue;   // hidden width > 2048: refused by hl_create
    return hipGetLastError();
  }
  const dim3 grid((maxRows + 3) / 4 + nEx);
  const int HQ = (a.H + 63) / 64;
  if (HQ <= 1) hipLaunchKernelGGL((head_kernel_t<1, 1>), grid, block, 0, s, a, ex);
  else hipLaunchKernelGGL((head_kernel_t<2, 1>), grid, block, 0, s, a, ex);
  return hipGetLastError();
}

}  // namespace hl
