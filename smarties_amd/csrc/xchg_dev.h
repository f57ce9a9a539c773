// smarties_amd/csrc/xchg_dev.h -- one chunk of a replica collective: what a workgroup of xchg_allreduce_kernel (xchg.hip) does, and --
// FOLD -- what the chunk workgroups at the end of the weight-gradient launch's grid do (gemm16.hip: dw_table_kernel), so that a
// replica's step is two launches instead of three.  The reference: MPI_Iallreduce of the gradient and AdamOptimizer::apply_update
// (Network/Optimizer.cpp:110-160), the counters' reduction (Utils/DelayedReductor.cpp:53-83).  Protocol: xchg.hip's header.
#pragma once
#include "tail_dev.h"

namespace hl {

// development time stamps of the folded launch (-DHL_FOLD_STAMPS; tools/fold_stamps.py): DevScalars::dbgT, 100 MHz
#ifdef HL_FOLD_STAMPS
#define FOSTAMP(sc_, i) do { if (threadIdx.x == 0) const_cast<DevScalars*>(sc_)->dbgT[i] = wall_clock64(); } while (0)
#else
#define FOSTAMP(sc_, i) do { } while (0)
#endif

template <typename T> struct Vec16 { T v[16 / sizeof(T)]; };

// (relaxed: the window is uncached memory, every load goes to HBM; an acquire load would invalidate this XCD's L2 at every poll)
__device__ __forceinline__ unsigned long long ldSys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void stSys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// a 16-byte unit of a window slot as two 8-byte system-scope loads (sc0 sc1: served by memory whatever an L2 may hold of the line)
template <typename V> __device__ __forceinline__ V ldWindowUnit(const V* p) {
  static_assert(sizeof(V) == 16, "16-byte units");
  union { unsigned long long q[2]; V v; } u;
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(p);
  u.q[0] = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); u.q[1] = __hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return u.v;
}

// what a chunk workgroup needs of a collective (kernel arguments of either launch; no record on the stack)
struct XchgCore {
  void* msg; long long n;                       // local message, summed in place
  int nRanks, rank;
  unsigned char* const* peers; size_t slotsOffset, slotBytes;
  XchgCtl* ctl; DevScalars* sc; long long timeoutTicks;
  long long pushed;                             // leading elements already in the peers' windows (FOLD: all of them)
  unsigned localTarget;                         // FOLD: arrivals (tiles + the bookkeeping rider of THIS launch) this replica's own push consists of
};
struct XchgAdam { float* W; float* M1; float* M2; long long n; float lambda, fac; int parity; };
// LDS of a chunk workgroup: 48 bytes, 8-byte aligned
struct XchgLds { unsigned long long seq; int last, fail; long long farDelta[2]; unsigned maxAbs, pad; };

// FUSE (the gradient message of a step): the workgroup that summed a chunk applies Adam to it and the last workgroup to finish runs the
// bookkeeping that needs the summed counters (MemoryProcessing::updateCounters ... beta, the next step's Adam scalars).
// FOLD (round 6; implies FUSE): the caller is a workgroup of the launch that PRODUCES the gradient.  Its tiles stored their values into
// every window -- the own one included: what this workgroup sums is read from the windows only, never from the tiles' cached stores,
// which another XCD's L2 may still hold -- and counted themselves on ctl->pushed once those stores were acknowledged; this workgroup
// waits for that count, then stamps the peers' flags and goes on as the separate exchange launch does.  Chunk workgroups sit at the
// END of the grid: every tile workgroup has been dispatched when the first of them starts, so they wait for running workgroups only.
template <typename T, bool FUSE, bool FOLD>
__device__ __forceinline__ void xchgChunk(const XchgCore& a, const XchgAdam& ad, const PostArgs& post, int postModeClose, int chunk, int nCh, XchgLds* L) {
  const int tid = threadIdx.x, R = a.nRanks, me = a.rank;
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 6);
  if (tid == 0) { L->seq = __hip_atomic_load(&a.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); L->fail = 0; }
  __syncthreads();
  const unsigned long long seq = L->seq, tag = seq + 1;
  const int par = (int)(seq & 1);
  // 16-byte units of the message; the last one may be partial (handled element-wise)
  const long long bytes = a.n * (long long)sizeof(T), full = bytes >> 4;
  const long long per = (full + nCh - 1) / nCh, v0 = per * chunk, v1 = min(full, v0 + per);
  typedef Vec16<T> V;
  V* msg = reinterpret_cast<V*>(a.msg);
  const size_t slotOff = a.slotsOffset + ((size_t)par * R + me) * a.slotBytes;
  const long long tail0 = full * (16 / (long long)sizeof(T));          // elements behind the last full unit: chunk 0 carries them
  if constexpr (FOLD) {
    // ---- this replica's own message is complete in every window once all of this launch's producers have arrived ----
    if (tid == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(&a.ctl->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tag) {      // (set by the last producer: foldArrive)
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > a.timeoutTicks) { __hip_atomic_store(&a.sc->errFlag, 79, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); L->fail = 1; break; }
      }
    }
    __syncthreads();
    if (chunk == 0) FOSTAMP(a.sc, 7);
  } else {
    // ---- push: this chunk into every peer's window (what the producing launch pushed itself -- the leading a.pushed elements of a
    // gradient message, PushArgs -- is already there: its stores were acknowledged before that launch ended) ----
    const long long vPushed = (a.pushed * (long long)sizeof(T)) >> 4;
    for (long long v = max(v0, vPushed) + tid; v < v1; v += 256) {
      const V x = msg[v];
      for (int p = 0; p < R; ++p) if (p != me) reinterpret_cast<V*>(a.peers[p] + slotOff)[v] = x;
    }
    if (chunk == 0 && tid < (int)(a.n - tail0)) {
      const T x = reinterpret_cast<const T*>(a.msg)[tail0 + tid];
      for (int p = 0; p < R; ++p) if (p != me) reinterpret_cast<T*>(a.peers[p] + slotOff)[tail0 + tid] = x;
    }
    __threadfence_system();
    __syncthreads();
  }
  unsigned long long* myFlags = reinterpret_cast<unsigned long long*>(a.peers[me]) + (size_t)par * R * XCHG_CHUNKS;
  if (tid < R && tid != me) {
    if (!FOLD || !L->fail) stSys(reinterpret_cast<unsigned long long*>(a.peers[tid]) + ((size_t)par * R + me) * XCHG_CHUNKS + chunk, tag);
    // ---- wait for the same chunk of every peer ----
    const unsigned long long* f = myFlags + (size_t)tid * XCHG_CHUNKS + chunk;
    const long long t0 = wall_clock64();
    while (ldSys(f) < tag) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeoutTicks) { __hip_atomic_store(&a.sc->errFlag, 79, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); L->fail = 1; break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);      // once, behind the last stamp
  }
  __syncthreads();
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 8);
  auto consensus = [&]() {
    if (tid == 0) {
      if (!L->fail) __hip_atomic_fetch_add(&a.ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      while (!L->fail && __hip_atomic_load(&a.ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nCh) {
        if (__hip_atomic_load(&a.sc->errFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { L->fail = 1; break; }
        if (wall_clock64() - t0 > 2 * a.timeoutTicks) { __hip_atomic_store(&a.sc->errFlag, 79, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); L->fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      // (ONE verdict per workgroup, taken here: the bookkeeping behind it holds barriers)
      if (__hip_atomic_load(&a.sc->errFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) L->fail = 1;
    }
    __syncthreads();
  };
  // ---- sum in rank order (round 6: in FRONT of the two-phase wait -- the sums change nothing but the message buffer, so they run under
  // the wait for the slowest chunk; four 16-byte units per thread and pass, all of their R loads in flight together: the windows are
  // uncached, a load is a round trip to HBM) ----
  const unsigned char* mine = a.peers[me] + a.slotsOffset + (size_t)par * R * a.slotBytes;
  constexpr int UB = 4, EPV = (int)(16 / sizeof(T));
  if (!L->fail) for (long long vb = v0 + tid; vb < v1; vb += 256 * UB) {
    V acc[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long long v = min(vb + 256ll * u, v1 - 1);      // (clamped: loads only, no effect)
      acc[u] = (!FOLD && me == 0) ? msg[v] : ldWindowUnit(reinterpret_cast<const V*>(mine) + v);
    }
    for (int r = 1; r < R; ++r) {
      V x[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const long long v = min(vb + 256ll * u, v1 - 1);
        x[u] = (!FOLD && r == me) ? msg[v] : ldWindowUnit(reinterpret_cast<const V*>(mine + (size_t)r * a.slotBytes) + v);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
#pragma unroll
        for (int q = 0; q < EPV; ++q) acc[u].v[q] += x[u].v[q];
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long long v = vb + 256ll * u;
      if (v < v1) {
        msg[v] = acc[u];
        if constexpr (FUSE) {
          // the units behind the parameters -- the summed counters -- are read by the workgroup that closes the step, on whichever XCD it
          // runs: they go to the coherence point themselves, so that no workgroup has to write back its XCD's L2 (the Adam results) for them
          if (v * EPV >= ad.n) {
#pragma unroll
            for (int q = 0; q < EPV; ++q) __hip_atomic_store(reinterpret_cast<T*>(a.msg) + v * EPV + q, acc[u].v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 9);
  // (Adam's first pass -- four units per thread: all of them on a node, where a 292 KB message has 64 chunks -- is COMPUTED in front of the
  //  two-phase wait, from this thread's own sums and operands that do not depend on the wait; only its stores stand behind it)
  constexpr int UA = 4;
  const long long vAdam = FUSE ? (ad.n + EPV - 1) / EPV : 0;      // units that hold parameters
  f32x4 w4[UA], m14[UA], m24[UA];
  AdamCoef c{};
  if constexpr (FUSE) {
    static_assert(sizeof(T) == 4, "Adam runs on float messages");
    c.eta = a.sc->etaEff[ad.parity]; c.lambda = ad.lambda; c.fac = ad.fac;
    f32x4 g4[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const long long v = max(0ll, min(min(v0 + tid + 256ll * u, v1 - 1), vAdam - 1));
      g4[u] = reinterpret_cast<const f32x4*>(a.msg)[v];      // (this thread's own store of a moment ago)
      w4[u] = reinterpret_cast<const f32x4*>(ad.W)[v]; m14[u] = reinterpret_cast<const f32x4*>(ad.M1)[v]; m24[u] = reinterpret_cast<const f32x4*>(ad.M2)[v];
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const long long v = v0 + tid + 256ll * u;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (v < v1 && v * 4 + q < ad.n) { float w = w4[u][q], m1 = m14[u][q], m2 = m24[u][q]; adamStep(c, g4[u][q], w, m1, m2); w4[u][q] = w; m14[u][q] = m1; m24[u][q] = m2; }
    }
  }
  if constexpr (FUSE) {
    // Two phases (round 5; ADVICE r03 / VERDICT r04): a chunk whose peers arrived used to sum and apply Adam at once -- if another
    // chunk then timed out, the parameter vector was left PARTIALLY updated.  Now every workgroup reports that its stamps came and
    // waits until all nCh have (they are resident together: at most XCHG_CHUNKS workgroups); a single failure -- the sticky device
    // error -- makes every workgroup skip its Adam slice: after error 79 weights and moments are those of before the collective (the
    // message buffer may hold sums: nothing reads it after a failure).  Costs one counter round trip among the launch's workgroups.
    consensus();
  }
  // A peer's message never came (or an earlier collective already failed: the error is sticky): no workgroup applies Adam or runs the
  // bookkeeping -- the parameters stay as they were (gradient messages: the two-phase wait above; the other messages have no side
  // effect beyond their own buffer).  The host sees HL_ERR_HIP at its next read-back.  The sequence still advances, so nothing waits
  // on this collective later.
  const bool failed = FUSE ? L->fail != 0 : (L->fail != 0 || __hip_atomic_load(&a.sc->errFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 15);
  if constexpr (FUSE) {
    // ---- Adam on the summed chunk: the first pass's results are stored, further passes (fewer, larger chunks: replicas sharing a device)
    // load, compute and store; parameters and moments as 16-byte accesses (the arrays are 16-byte aligned and hold ad.n rounded up to four
    // elements: Parameters.h layout + PARAM_TAIL slack)
    if (!failed) {
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const long long v = v0 + tid + 256ll * u;
        if (v < v1 && v < vAdam) { reinterpret_cast<f32x4*>(ad.W)[v] = w4[u]; reinterpret_cast<f32x4*>(ad.M1)[v] = m14[u]; reinterpret_cast<f32x4*>(ad.M2)[v] = m24[u]; }
      }
    }
    // ---- the step's closing bookkeeping (MemoryProcessing::updateCounters ... beta, the next step's Adam scalars) by the workgroup that
    // summed the END of the message -- the counters: its own stores -- as soon as it knows that nobody failed, beside the other workgroups'
    // Adam stores and -- it is one thread's work behind one barrier -- beside this workgroup's own later passes (until round 6: by the last
    // workgroup to finish, behind everything).  It touches scalars no Adam slice reads
    // (etaEff of the OTHER buffer slot), so nothing has to wait for it but the end of the launch.
    if (chunk == nCh - 1 && !failed) {
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (the bookkeeping rider's scalars, where it ran in this launch: an invalidate, no write-back)
      if (FOLD) FOSTAMP(a.sc, 12);
      postPart(post, L->farDelta, &L->maxAbs, nullptr, 0, FOLD ? postModeClose : -1);
      if (FOLD) FOSTAMP(a.sc, 13);
    }
    if (!failed) for (long long vb = v0 + tid + 256ll * UA; vb < v1 && vb < vAdam; vb += 256 * UA) {
      f32x4 g4[UA];
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const long long v = min(min(vb + 256ll * u, v1 - 1), vAdam - 1);
        g4[u] = reinterpret_cast<const f32x4*>(a.msg)[v];
        w4[u] = reinterpret_cast<const f32x4*>(ad.W)[v]; m14[u] = reinterpret_cast<const f32x4*>(ad.M1)[v]; m24[u] = reinterpret_cast<const f32x4*>(ad.M2)[v];
      }
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const long long v = vb + 256ll * u;
        if (v < v1 && v < vAdam) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (v * 4 + q < ad.n) { float w = w4[u][q], m1 = m14[u][q], m2 = m24[u][q]; adamStep(c, g4[u][q], w, m1, m2); w4[u][q] = w; m14[u][q] = m1; m24[u][q] = m2; }
          reinterpret_cast<f32x4*>(ad.W)[v] = w4[u]; reinterpret_cast<f32x4*>(ad.M1)[v] = m14[u]; reinterpret_cast<f32x4*>(ad.M2)[v] = m24[u];
        }
      }
    }
  }
  if (!FOLD && !failed && chunk == 0 && tid < (int)(a.n - tail0)) {      // (FOLD: the message is a whole number of 16-byte units, checked by the host)
    T acc = 0;
    for (int r = 0; r < R; ++r) {
      const T x = r == me ? reinterpret_cast<const T*>(a.msg)[tail0 + tid] : reinterpret_cast<const T*>(mine + (size_t)r * a.slotBytes)[tail0 + tid];
      acc = r == 0 ? x : acc + x;
    }
    reinterpret_cast<T*>(a.msg)[tail0 + tid] = acc;
  }
  // ---- the last workgroup to get here closes the collective: every workgroup has read `seq` by then ----
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 10);
  // (FUSE: what another workgroup of this launch reads of this one's results -- the summed counters -- went out by itself above; the
  //  parameters are read by later launches only.  Other messages: their sums are released here as before.)
  if constexpr (!FUSE) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (FOLD && chunk == 0) FOSTAMP(a.sc, 11);
  if (tid == 0) {
    const bool last = atomicAdd(&a.ctl->done, 1u) == (unsigned)nCh - 1;
    L->last = last ? 1 : 0;
    if (last) {
      a.ctl->done = 0; a.ctl->arrived = 0;      // (every workgroup left the two-phase wait before it added to `done`)
      __hip_atomic_store(&a.ctl->seq, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the arrival of one producer of a folded launch (a tile workgroup, the bookkeeping rider): every wavefront's window stores are
// acknowledged, then one count; the LAST of the launch's `target` producers re-arms the counter and publishes `ready` (all counts in
// front of its own were taken behind acknowledged stores), which is what the chunk workgroups poll -- on a cache line of its own
__device__ __forceinline__ void foldArrive(XchgCtl* ctl, unsigned target) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&ctl->pushed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == target) {
#ifdef HL_FOLD_STAMPS
      ctl->pad3[0] = (unsigned long long)wall_clock64();
#endif
      __hip_atomic_store(&ctl->pushed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long seq = __hip_atomic_load(&ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl->ready, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace hl
