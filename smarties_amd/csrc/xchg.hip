// smarties_amd/csrc/xchg.hip -- the replicas' sum over xGMI as ONE kernel (the reference: MPI_Iallreduce of the gradient,
// Network/Optimizer.cpp:110-132; of the counters and moments, Utils/DelayedReductor.cpp:53-83).
//
// A 292 KB message on 8 GPUs is latency: a ring or tree all-reduce pays its hop count, the direct xGMI links of a node let every
// replica WRITE its message straight into a window in each peer's HBM instead.  Per replica one window (uncached device memory,
// peer-mapped through hipIpc handles or, inside one process, plain pointers):
//     flags [2][nRanks][XCHG_CHUNKS]   64-bit arrival stamps (sequence number + 1) per sender and chunk of the message
//     slots [2][nRanks][slotBytes]     the senders' messages
// double-buffered over the parity of the sequence number: a replica can be at most one collective ahead of the slowest one
// (it needs everybody's message of collective s before it can issue s + 1), so the buffer of s - 1 is free when s + 1 writes it.
// Workgroup c owns chunk c of the message: it stores its chunk into all peers' windows, fences at system scope, stamps the
// peers' flags, waits for the peers' stamps on ITS chunk only (no device-wide barrier), and sums the nRanks contributions in
// rank order -- the own one from the local buffer -- so that every replica gets the same bits.  The waits are bounded: a lost
// peer raises the learner's sticky device error (hl_sync then returns HL_ERR_HIP) instead of hanging the GPU.
#include "xchg_dev.h"

namespace hl {

// (the chunk's work: xchg_dev.h -- shared with the chunk workgroups of the folded weight-gradient launch, gemm16.hip)
template <typename T, bool FUSE>
__global__ __launch_bounds__(256) void xchg_allreduce_kernel(XchgArgs a) {
  __shared__ XchgLds L;
  XchgCore c; c.msg = a.msg; c.n = a.n; c.nRanks = a.nRanks; c.rank = a.rank; c.peers = a.peers; c.slotsOffset = a.slotsOffset; c.slotBytes = a.slotBytes;
  c.ctl = a.ctl; c.sc = a.sc; c.timeoutTicks = a.timeoutTicks; c.pushed = a.pushed; c.localTarget = 0u;
  XchgAdam ad; ad.W = a.adam.W; ad.M1 = a.adam.M1; ad.M2 = a.adam.M2; ad.n = a.adam.n; ad.lambda = a.adam.lambda; ad.fac = a.adam.fac; ad.parity = a.adam.parity;
  xchgChunk<T, FUSE, false>(c, ad, a.post, 0, (int)blockIdx.x, (int)gridDim.x, &L);
}

__global__ __launch_bounds__(256) void xchg_clean_kernel(unsigned char* win, size_t slotsOffset, size_t slotBytes, int nRanks, const XchgCtl* ctl, long long bytes) {
  const unsigned long long seq = __hip_atomic_load(&ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (the collective just closed was seq - 1)
  const int par = (int)((seq - 1) & 1);
  const long long n16 = (bytes + 15) >> 4;
  for (int r = blockIdx.y; r < nRanks; r += gridDim.y) {
    u32x4* q = reinterpret_cast<u32x4*>(win + slotsOffset + ((size_t)par * nRanks + r) * slotBytes);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) q[i] = u32x4{0u, 0u, 0u, 0u};
  }
}
hipError_t launch_xchg_clean(unsigned char* win, size_t slotsOffset, size_t slotBytes, int nRanks, const XchgCtl* ctl, long long bytes, hipStream_t s) {
  const long long n16 = (bytes + 15) >> 4;
  int bx = (int)((n16 + 255) / 256); if (bx > 64) bx = 64; if (bx < 1) bx = 1;
  hipLaunchKernelGGL(xchg_clean_kernel, dim3(bx, nRanks), dim3(256), 0, s, win, slotsOffset, slotBytes, nRanks, ctl, bytes);
  return hipGetLastError();
}

int xchg_chunks(long long bytes, int maxChunks) {
  const int cap = maxChunks >= 1 && maxChunks <= XCHG_CHUNKS ? maxChunks : XCHG_CHUNKS;
  int nCh = (int)((bytes + 4095) / 4096); if (nCh < 1) nCh = 1; if (nCh > cap) nCh = cap;
  return nCh;
}
hipError_t launch_xchg_allreduce(const XchgArgs& a, int dtype, hipStream_t s) {
  const long long bytes = a.n * (dtype == 0 ? 4 : 8);
  const int nCh = xchg_chunks(bytes, a.maxChunks);
  if (a.fuse && dtype == 0) hipLaunchKernelGGL((xchg_allreduce_kernel<float, true>), dim3(nCh), dim3(256), 0, s, a);
  else if (dtype == 0) hipLaunchKernelGGL((xchg_allreduce_kernel<float, false>), dim3(nCh), dim3(256), 0, s, a);
  else if (dtype == 1) hipLaunchKernelGGL((xchg_allreduce_kernel<double, false>), dim3(nCh), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((xchg_allreduce_kernel<long long, false>), dim3(nCh), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace hl
