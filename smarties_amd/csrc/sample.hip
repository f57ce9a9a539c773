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
// ---------------------------------------------------------------------------------------------------------------------
// Large local batches (1024 < B <= 16384): the same sampler -- Sample_uniform::sample + IDtoSeqStep (Sampling.cpp:26-47,82-96):
// Lemire draws from the learner's mt19937 consumed in order, sort, unique, redraw of the missing ones, the Adam draws, then flat
// index -> (episode, step) and the truncated next states' rows -- as ONE workgroup of 1024 threads with the candidates in LDS
// (64 KB at 16384).  The words are drawn 1024 at a time, never more than are still needed, so the generator ends where the
// sequential algorithm leaves it; compactions run over chunks of 1024 elements in order.  No in-kernel gather (the states are
// assembled by stack_gather_kernel), no riders: these batches step eagerly or as graphs with the sampler as a branch.
// ---------------------------------------------------------------------------------------------------------------------
#define BIG_NT 1024
#define BIG_MAXB 16384
// development time stamps of the sampler's phases, 100 MHz clock: -DHL_BIGSAMPLE_STAMPS (tools/bigsample_stamps.py)
#ifdef HL_BIGSAMPLE_STAMPS
#define BSTMP(i) do { if (threadIdx.x == 0) sc->dbgT[i] = wall_clock64(); } while (0)
#else
#define BSTMP(i) do { } while (0)
#endif
__device__ void bigTwist(unsigned* x, unsigned* xo) {
  const int tid = threadIdx.x;
  if (tid < 624) xo[tid] = x[tid];
  __syncthreads();
  if (tid < 227) x[tid] = xo[tid + 397] ^ mtF(xo[tid], xo[tid + 1]);
  __syncthreads();
  if (tid < 227) x[227 + tid] = x[tid] ^ mtF(xo[227 + tid], xo[228 + tid]);
  __syncthreads();
  if (tid < 169) x[454 + tid] = x[227 + tid] ^ mtF(xo[454 + tid], xo[455 + tid]);
  __syncthreads();
  if (tid == 0) x[623] = x[396] ^ mtF(xo[623], x[0]);
  __syncthreads();
}
__device__ void bigDraw(unsigned* x, unsigned* xo, int* pPos, unsigned* raw, int n) {      // n <= BIG_NT tempered words into raw[0..n)
  int done = 0;
  while (done < n) {
    int pos = *pPos;
    __syncthreads();
    if (pos >= 624) { bigTwist(x, xo); pos = 0; }
    const int take = min(n - done, 624 - pos);
    if ((int)threadIdx.x < take) raw[done + threadIdx.x] = mtTemper(x[pos + threadIdx.x]);
    if (threadIdx.x == 0) *pPos = pos + take;
    __syncthreads();
    done += take;
  }
}
// exclusive scan of one flag per thread over the workgroup's 16 wavefronts; returns the total (two barriers)
__device__ int bigScan(bool flag, int* excl, int* sWave /*[16]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long m = __ballot(flag);
  if (lane == 0) sWave[wave] = __popcll(m);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < BIG_NT / 64; ++w) { const int c = sWave[w]; total += c; if (w < wave) before += c; }
  *excl = before + __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  return total;
}
#define BIG_Q (BIG_MAXB / BIG_NT)
// the same scan for BIG_Q flags per thread at once (bit q of `flags`: element q * 1024 + tid, elements in index order): exclusive ranks in
// ex[], returns the total; three barriers instead of two per 1024 elements.  sCnt: [BIG_Q * 16 + 4]
__device__ int bigScanAll(unsigned flags, int* ex, int* sCnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long m[BIG_Q];
#pragma unroll
  for (int q = 0; q < BIG_Q; ++q) { m[q] = __ballot((flags >> q) & 1u); if (lane == 0) sCnt[q * 16 + wave] = __popcll(m[q]); }
  __syncthreads();
  int c = 0, inc = 0;
  if (tid < BIG_Q * 16) {
    c = sCnt[tid]; inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) sCnt[BIG_Q * 16 + wave] = inc;
  }
  __syncthreads();
  if (tid < BIG_Q * 16) {
    int before = 0;
    for (int w = 0; w < wave; ++w) before += sCnt[BIG_Q * 16 + w];
    sCnt[tid] = before + inc - c;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < BIG_Q; ++q) ex[q] = sCnt[q * 16 + wave] + __popcll(m[q] & lt);
  int total = 0;
#pragma unroll
  for (int w = 0; w < BIG_Q * 16 / 64; ++w) total += sCnt[BIG_Q * 16 + w];
  __syncthreads();
  return total;
}
// vals[0..B) sorted: keeps the first of every run of equal values, in order; returns how many remain.  Every element is read before
// the scan's barriers, written behind them.
__device__ int bigUnique(unsigned* vals, int* sCnt, int B) {
  const int tid = threadIdx.x;
  unsigned v[BIG_Q]; unsigned flags = 0u;
#pragma unroll
  for (int q = 0; q < BIG_Q; ++q) {
    const int i = tid + BIG_NT * q;
    v[q] = i < B ? vals[i] : 0u;
    if (i < B && (i == 0 || v[q] != vals[i - 1])) flags |= 1u << q;
  }
  int ex[BIG_Q];
  const int total = bigScanAll(flags, ex, sCnt);
#pragma unroll
  for (int q = 0; q < BIG_Q; ++q) if ((flags >> q) & 1u) vals[ex[q]] = v[q];
  __syncthreads();
  return total;
}
__device__ void bigDrawAccepted(unsigned* x, unsigned* xo, int* sPos, unsigned* raw, unsigned* vals, int* sWave, int from, int B, unsigned range, unsigned threshold) {
  int filled = from;
  while (filled < B) {
    const int need = min(B - filled, BIG_NT);
    bigDraw(x, xo, sPos, raw, need);
    bool fl = false; unsigned v = 0;
    if ((int)threadIdx.x < need) {
      const unsigned long long prod = (unsigned long long)raw[threadIdx.x] * (unsigned long long)range;
      fl = (unsigned)prod >= threshold; v = (unsigned)(prod >> 32);
    }
    int ex; const int acc = bigScan(fl, &ex, sWave);
    if (fl) vals[filled + ex] = v;
    filled += acc;
  }
  __syncthreads();
}
// the prioritised samplers' draws (tail_dev.h: drawPER -- std::discrete_distribution over the cumulative table, for PERseq a step inside
// the drawn episode): BIG_NT / words-per-value values per round, every value consuming its two (three) generator words in order
__device__ void bigDrawPER(const SampleArgs& a, unsigned* x, unsigned* xo, int* sPos, unsigned* raw, unsigned* vals, int from, int B) {
  const int tid = threadIdx.x, Wn = a.perAlgo == HL_SAMPLE_PERSEQ ? 3 : 2, per = BIG_NT / Wn;
  for (int c0 = from; c0 < B; c0 += per) {
    const int n = min(per, B - c0);
    bigDraw(x, xo, sPos, raw, n * Wn);
    if (tid < n) {
      const unsigned w0 = raw[Wn * tid], w1 = raw[Wn * tid + 1];
      double p = ((double)w0 + (double)w1 * 4294967296.0) / 18446744073709551616.0;
      if (p >= 1.0) p = 0.99999999999999988898;                  // nextafter(1, 0)
      long long lo = 0;
      if (a.perN >= 2) {                                          // std::lower_bound: first entry not less than p
        long long len = a.perN;
        while (len > 0) { const long long half = len >> 1; if (a.perCp[lo + half] < p) { lo += half + 1; len -= half + 1; } else len = half; }
      }
      unsigned v = (unsigned)lo;
      if (a.perAlgo == HL_SAMPLE_PERSEQ) {
        float u = (float)raw[Wn * tid + 2] / 4294967296.0f;
        if (u >= 1.0f) u = 0.99999994f;                           // nextafterf(1, 0)
        const PosRec rec = a.rp.posRec[lo];
        v = (unsigned)(rec.prefix + (long long)(unsigned long long)(u * (float)(unsigned long long)(rec.N - 1)));
      }
      vals[c0 + tid] = v;
    }
    __syncthreads();
  }
  __syncthreads();
}
__device__ void bigBitonicSort(unsigned* vals, int Bp) {      // Bp a power of two >= 2048
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto waveRounds = [&](int k, int jStart) {                 // strides below 64: inside the wavefronts, each owning the 64-element blocks wave, wave + 16, ...
    for (int blk = wave; blk * 64 < Bp; blk += BIG_NT / 64) {
      const int i = blk * 64 + lane;
      unsigned key = vals[i];
      if (k <= 64) {
        for (int kk = 2; kk <= k; kk <<= 1)
          for (int j = kk >> 1; j > 0; j >>= 1)
            key = cmpx(key, (unsigned)__shfl_xor((int)key, j, 64), (i & j) == 0, (i & kk) == 0);
      } else {
        for (int j = jStart; j > 0; j >>= 1)
          key = cmpx(key, (unsigned)__shfl_xor((int)key, j, 64), (i & j) == 0, (i & k) == 0);
      }
      vals[i] = key;
    }
  };
  __syncthreads();
  waveRounds(64, 32);
  for (int k = 128; k <= Bp; k <<= 1) {
    for (int j = k >> 1; j >= 64; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < Bp / 2; t += BIG_NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const unsigned a = vals[i], b = vals[i | j];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { vals[i] = b; vals[i | j] = a; }
      }
    }
    __syncthreads();
    waveRounds(k, 32);
  }
  __syncthreads();
}
// The draws are uniform over [0, range): 4096 buckets by v * 4096 / range (monotone in v), counted and scattered with LDS atomics (the
// elements of a bucket then stand together, in any order); every element then counts the elements of its bucket in front of it -- four
// on average at 16384, read four at a time; a wavefront's 64 neighbours share a few buckets, so the lanes of a read mostly hit the same
// addresses -- and goes to its place.  The sorting network over 16384 elements took ~125 of the sampler's 258 us, this 12.  A bucket
// beyond 64 elements (a replay of a few thousand transitions: most draws collide) or a range below 2^16 leaves vals untouched and
// returns false: the network then.
#define BIG_NB 4096
__device__ bool bigBucketSort(unsigned* vals, unsigned* tmp /*[B + 3]*/, int* cnt /*[BIG_NB]*/, int* sWave, int B, unsigned range, DevScalars* sc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (range < (1u << 16)) return false;
  const unsigned scale = (unsigned)(((unsigned long long)BIG_NB << 32) / range);      // v * scale >> 32 < 4096 for v < range
  BSTMP(8);
  int4* cnt4 = reinterpret_cast<int4*>(cnt);
  cnt4[tid] = make_int4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < B; i += BIG_NT) atomicAdd(&cnt[__umulhi(vals[i], scale)], 1);
  __syncthreads();
  BSTMP(9);
  const int4 c = cnt4[tid];                      // the thread's four buckets
  const int csum = (c.x + c.y) + (c.z + c.w);
  if (__syncthreads_or(c.x > 64 || c.y > 64 || c.z > 64 || c.w > 64)) return false;
  int inc = csum;                                // inclusive prefix over the workgroup
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
  if (lane == 63) sWave[wave] = inc;
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int w = 0; w < BIG_NT / 64; ++w) if (w < wave) before += sWave[w];
  const int st = before + inc - csum;
  cnt4[tid] = make_int4(st, st + c.x, st + c.x + c.y, st + c.x + c.y + c.z);
  __syncthreads();
  BSTMP(10);
  for (int i = tid; i < B; i += BIG_NT) { const unsigned v = vals[i]; tmp[atomicAdd(&cnt[__umulhi(v, scale)], 1)] = v; }
  __syncthreads();                               // cnt[b] is now the end of bucket b
  BSTMP(11);
#pragma unroll 2
  for (int i = tid; i < B; i += BIG_NT) {
    const unsigned v = tmp[i];
    const int b = (int)__umulhi(v, scale), s0 = b ? cnt[b - 1] : 0, e0 = cnt[b];
    int r = 0;
    for (int j = s0; j < e0; j += 4) {           // (reads up to three words behind the bucket: inside tmp or the scan's scratch behind it)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int jj = j + k; const unsigned u = tmp[jj]; r += (jj < e0 && (u < v || (u == v && jj < i))) ? 1 : 0; }      // (equal values: in the order they stand)
    }
    vals[s0 + r] = v;
  }
  __syncthreads();
  BSTMP(12);
  return true;
}
__device__ int bigSortUnique(unsigned* vals, unsigned* tmp, int* cnt, int* sWave, int B, int Bp, unsigned range, DevScalars* sc) {
  if (!bigBucketSort(vals, tmp, cnt, sWave, B, range, sc)) {
    for (int i = B + threadIdx.x; i < Bp; i += BIG_NT) vals[i] = 0xFFFFFFFFu;
    bigBitonicSort(vals, Bp);
  }
  return bigUnique(vals, sWave, B);
}
// Redraw rounds: vals[0..have) is sorted and unique, vals[have..have + m) holds the m <= BIG_TAIL new draws.  Sorting everything
// again costs ~100 us at 16384; instead the tail is sorted on its own (in sT), every element finds its rank in the other sequence by
// binary search (head elements first among equals), and one scatter merges the two; then the unique pass as above.
#define BIG_TAIL 4096
__device__ int bigMergeUnique(unsigned* vals, unsigned* sT, int* sWave, int have, int m) {
  const int tid = threadIdx.x;
  int P2 = 64; while (P2 < m) P2 <<= 1;
  for (int i = tid; i < P2; i += BIG_NT) sT[i] = i < m ? vals[have + i] : 0xFFFFFFFFu;
  bigBitonicSort(sT, P2);
  unsigned hv[BIG_MAXB / BIG_NT], tv[BIG_TAIL / BIG_NT]; int hd[BIG_MAXB / BIG_NT], td[BIG_TAIL / BIG_NT];
  // head elements: how many of the tail are smaller (the tail stands padded with 0xFFFFFFFF to P2 = 2^k, above every draw): log2(P2) steps
  // of a fixed-length search, the thread's sixteen searches side by side
#pragma unroll
  for (int q = 0; q < BIG_MAXB / BIG_NT; ++q) { const int i = tid + BIG_NT * q; hv[q] = i < have ? vals[i] : 0u; hd[q] = 0; }
  for (int hs = P2 >> 1; hs >= 1; hs >>= 1) {
#pragma unroll
    for (int q = 0; q < BIG_MAXB / BIG_NT; ++q) if (sT[hd[q] + hs - 1] < hv[q]) hd[q] += hs;
  }
#pragma unroll
  for (int q = 0; q < BIG_MAXB / BIG_NT; ++q) { const int i = tid + BIG_NT * q; if (sT[hd[q]] < hv[q]) hd[q] += 1; hd[q] = i < have ? i + hd[q] : -1; }      // (the last element: a tail of exactly P2 draws has no padding)
#pragma unroll
  for (int q = 0; q < BIG_TAIL / BIG_NT; ++q) {
    const int j = tid + BIG_NT * q;
    td[q] = -1; tv[q] = 0u;
    if (j < m) {
      const unsigned v = sT[j];
      int lo = 0, len = have;                    // head elements not larger than v
      while (len > 0) { const int half = len >> 1; if (vals[lo + half] <= v) { lo += half + 1; len -= half + 1; } else len = half; }
      tv[q] = v; td[q] = j + lo;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < BIG_MAXB / BIG_NT; ++q) if (hd[q] >= 0) vals[hd[q]] = hv[q];
#pragma unroll
  for (int q = 0; q < BIG_TAIL / BIG_NT; ++q) if (td[q] >= 0) vals[td[q]] = tv[q];
  __syncthreads();
  return bigUnique(vals, sWave, have + m);
}
__global__ __launch_bounds__(BIG_NT) void big_sample_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* vals = reinterpret_cast<unsigned*>(smem);               // [Bp]
  int Bp = 2048; while (Bp < a.B) Bp <<= 1;
  unsigned* x = vals + Bp; unsigned* xo = x + 624; unsigned* raw = xo + 624;      // [624] [624] [BIG_NT]
  unsigned* sT = raw + BIG_NT;                                                    // [BIG_TAIL] the redrawn tail, sorted
  unsigned* tmp = sT + BIG_TAIL;                                                  // [Bp] the bucket sort's scatter target (its counters: raw)
  int* sWave = reinterpret_cast<int*>(tmp + Bp); int* sPos = sWave + BIG_Q * 16 + 4;      // sWave: [BIG_Q * 16 + 4] (bigScanAll)
  const int tid = threadIdx.x, B = a.B;
  DevScalars* sc = a.sc;
  if (tid < 624) { const unsigned v = sc->rng[tid]; x[tid] = v; if (a.backupRng) sc->rngBak[tid] = v; }      // (a minibatch drawn ahead may be discarded: dropPresample)
  if (tid == 0) { const unsigned p0 = sc->rngPos; *sPos = (int)p0; if (a.backupRng) sc->rngBakPos = p0; sc->sampleSeq += 1; }
  const unsigned long long nData = (unsigned long long)sc->nTransitions;
  const int nEp = (int)sc->nEpisodes;
  const unsigned range = (unsigned)nData;
  const unsigned threshold = range ? (0u - range) % range : 0u;
  __syncthreads();
  BSTMP(0);
  if (a.flatGiven) { for (int i = tid; i < B; i += BIG_NT) vals[i] = (unsigned)a.flatGiven[i]; __syncthreads(); }
  else {
    if (a.perAlgo) bigDrawPER(a, x, xo, sPos, raw, vals, 0, B);
    else bigDrawAccepted(x, xo, sPos, raw, vals, sWave, 0, B, range, threshold);
    BSTMP(1);
    int have = bigSortUnique(vals, tmp, reinterpret_cast<int*>(sT), sWave, B, Bp, range, sc);
    BSTMP(2);
    while (have < B) {                       // duplicates: redraw the missing ones (Sampling.cpp:86-93)
      if (a.perAlgo) bigDrawPER(a, x, xo, sPos, raw, vals, have, B);
      else bigDrawAccepted(x, xo, sPos, raw, vals, sWave, have, B, range, threshold);
      have = B - have <= BIG_TAIL ? bigMergeUnique(vals, sT, sWave, have, B - have) : bigSortUnique(vals, tmp, reinterpret_cast<int*>(sT), sWave, B, Bp, range, sc);
    }
  }
  BSTMP(3);
  for (int d = 0; d < a.adamDraws; d += BIG_NT) bigDraw(x, xo, sPos, raw, min(a.adamDraws - d, BIG_NT));      // AdamOptimizer::apply_update (Optimizer.cpp:139)
  BSTMP(4);
  if (tid < 624) sc->rng[tid] = x[tid];
  if (tid == 0) sc->rngPos = (unsigned)*sPos;
  // ---- IDtoSeqStep + rows of the truncated next states, in minibatch order: four elements' table records requested at once (the guess
  //      of findPosition, tail_dev.h; the search behind it on a miss), one scan over all the next-state flags ----
  unsigned nf = 0u;
  for (int q0 = 0; q0 < BIG_Q; q0 += 4) {
    if (q0 * BIG_NT >= B) break;
    PosRec rec[4]; long long p1[4], f[4]; int k0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = (q0 + u) * BIG_NT + tid;
      f[u] = b < B ? (long long)vals[b] : 0ll;
      int k = (int)(((double)f[u] * (double)nEp) / (double)nData);
      k0[u] = min(max(k, 0), nEp - 1);
      rec[u] = a.rp.posRec[k0[u]]; p1[u] = a.rp.posRec[k0[u] + 1].prefix;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = (q0 + u) * BIG_NT + tid;
      if (b < B) {
        int lo = k0[u];
        if (f[u] < rec[u].prefix || f[u] >= p1[u]) {
          int l = f[u] < rec[u].prefix ? 0 : k0[u] + 1, hgh = f[u] < rec[u].prefix ? k0[u] : nEp;   // largest k in [l,hgh): prefix[k] <= f
          while (hgh - l > 1) { const int mid = (l + hgh) >> 1; if (a.rp.posRec[mid].prefix <= f[u]) l = mid; else hgh = mid; }
          lo = l; rec[u] = a.rp.posRec[lo];
        }
        const int t = (int)(f[u] - rec[u].prefix);
        a.bt.flat[b] = f[u]; a.bt.pos[b] = lo; a.bt.eid[b] = rec[u].eidTerm & 0x7fffffff; a.bt.t[b] = t; a.bt.tag[b] = rec[u].tag; a.bt.slot[b] = rec[u].off + t;
        if (t + 2 == rec[u].N && rec[u].eidTerm >= 0) nf |= 1u << (q0 + u);      // Episode::isTruncated(t+1) (Episode.h:158-161)
      }
    }
  }
  int exn[BIG_Q];
  const int base = bigScanAll(nf, exn, sWave);
#pragma unroll
  for (int q = 0; q < BIG_Q; ++q) {
    const int b = q * BIG_NT + tid;
    if (b < B) { const bool hn = (nf >> q) & 1u; a.bt.nextOf[b] = hn ? B + exn[q] : -1; if (hn) a.bt.nextSrc[exn[q]] = b; }
  }
  BSTMP(5);
  if (tid == 0) {
    sc->nNext[a.parity] = base; sc->nRows[a.parity] = B + base;
    if (a.computeEta) sc->etaEff[a.parity] = adamEtaEff(sc->nStep, sc->adam_bt1, sc->adam_bt2, a.eta0, a.epsAnneal);
  }
}
hipError_t launch_big_sample(const SampleArgs& a, hipStream_t s) {
  if (a.B > BIG_MAXB || !a.noGather) return hipErrorInvalidValue;
  int Bp = 2048; while (Bp < a.B) Bp <<= 1;
  const size_t lds = (size_t)4 * (2 * Bp + 624 + 624 + BIG_NT + BIG_TAIL) + 4 * (BIG_Q * 16 + 4 + 4);
  static size_t have = 0;
  if (lds > have) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(big_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; have = lds; }
  hipLaunchKernelGGL(big_sample_kernel, dim3(1), dim3(BIG_NT), lds, s, a);
  return hipGetLastError();
}
// the episode records of a large minibatch: one workgroup per 256 samples (the rest of the bookkeeping pass follows as its own launch)
__global__ __launch_bounds__(256) void post_agg_chunks_kernel(PostArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64];
  postPart(a, reinterpret_cast<long long*>(smem), reinterpret_cast<unsigned*>(smem + 8));
}
hipError_t launch_post_agg_chunks(const PostArgs& a, hipStream_t s) {
  PostArgs c = a; c.aggChunk = 1; c.mode = POST_AGG;
  hipLaunchKernelGGL(post_agg_chunks_kernel, dim3((a.B + 255) / 256), dim3(256), 0, s, c);
  return hipGetLastError();
}
hipError_t launch_sample(const SampleArgs& a, hipStream_t s) { return a.B > SMAXB ? launch_big_sample(a, s) : launch_step_tail(nullptr, &a, s); }
hipError_t launch_post(const PostArgs& a, hipStream_t s) { return launch_step_tail(&a, nullptr, s); }

}  // namespace hl
