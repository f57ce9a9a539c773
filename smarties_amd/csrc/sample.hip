// smarties_amd/csrc/sample.hip -- stand-alone launches of the step tail (tail_dev.h): the whole
// sampler and/or the bookkeeping pass as their own two-workgroup kernel (eager path, first
// minibatch of a replayed graph, parity tests).
#include "tail_dev.h"

namespace hl {

struct TailArgs { PostArgs post; SampleArgs samp; int doPost, doSample, phases; };

// workgroup 0 samples when both parts are requested (or alone); the other one does the bookkeeping
__global__ __launch_bounds__(256) void step_tail_kernel(TailArgs ta) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TAIL_LDS_BYTES];
  const bool sampler = ta.doSample && (!ta.doPost || blockIdx.x == 0);
  if (sampler) samplePhases(ta.samp, ta.phases, smem);
  else if (ta.doPost) postPhase(ta.post, smem);
}

hipError_t launch_step_tail(const PostArgs* post, const SampleArgs* samp, hipStream_t s, int phases) {
  TailArgs ta{};
  ta.doPost = post != nullptr; ta.doSample = samp != nullptr; ta.phases = phases;
  if (post) ta.post = *post;
  if (samp) ta.samp = *samp;
  const int blocks = (post && samp) ? 2 : 1;
  hipLaunchKernelGGL(step_tail_kernel, dim3(blocks), dim3(256), 0, s, ta);
  return hipGetLastError();
}
hipError_t launch_sample(const SampleArgs& a, hipStream_t s) { return launch_step_tail(nullptr, &a, s); }
hipError_t launch_post(const PostArgs& a, hipStream_t s) { return launch_step_tail(&a, nullptr, s); }

}  // namespace hl
