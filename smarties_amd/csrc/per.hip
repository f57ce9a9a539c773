// smarties_amd/csrc/per.hip -- the prioritised samplers' distribution (dataSamplingAlgo PERrank / PERerr / PERseq;
// ReplayMemory/Sampling.cpp:101-296): before every minibatch the reference rebuilds a std::discrete_distribution<Uint>
// over all stored transitions (episodes).  libstdc++ normalises it with a SEQUENTIAL double-precision accumulate, divides,
// and builds the cumulative table with a SEQUENTIAL partial_sum; the drawn indices depend on every rounding of those two
// chains.  Round 4: the chains are no longer WALKED -- inside a binade of the accumulator a round-to-nearest-even fp64 addition
// is an integer step, so the table is an integer prefix sum with real additions only at binade crossings and exact ties
// (per_scan_kernel below: bit-identical to the walk, 2.4 ms instead of 23 ms per million transitions in one workgroup, 0.55 ms
// in the grid form behind it; the reference's CPU takes ~4 ms).  (The samplers are not selected by any shipped settings file, and the importance weights
// they define are not applied in this version of the reference, Approximator.h:196).  The draws themselves
// (generate_canonical<double, 53> + lower_bound, sort / unique / redraw) run in the sampler kernel (tail_dev.h).
#include "dev_common.h"
#include <hipcub/hipcub.hpp>

namespace hl {

// probabilities in table order: flat index = prefix of the episode + step (PERerr, PERrank keys), or one per episode (PERseq)
__global__ __launch_bounds__(256) void per_probs_kernel(PerArgs a) {
  const float EPS = FLT_EPSILON;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = blockIdx.x * 4 + wave; p < a.nEpisodes; p += gridDim.x * 4) {
    const PosRec rec = a.rp.posRec[p];
    const int e = rec.eidTerm & 0x7fffffff, nd = rec.N - 1;
    if (a.algo == HL_SAMPLE_PERSEQ) {       // Sample_impSeq::prepare (:247-249): (avg squared error + eps)^(1/4) x length
      if (lane == 0) a.prob[p] = sqrtf(sqrtf(a.rp.epAgg[(size_t)e * AGG_N + AGG_AVGSQERR] + EPS)) * (float)(unsigned long long)nd;
      continue;
    }
    for (int j = lane; j < nd; j += 64) {
      const float dq = a.rp.DQ[rec.off + j], d2 = dq * dq;
      if (a.algo == HL_SAMPLE_PERERR) a.prob[rec.prefix + j] = sqrtf(sqrtf(d2 + EPS));       // TSample_impErr::prepare (:192-196)
      else { a.key[rec.prefix + j] = d2; a.idx[rec.prefix + j] = (unsigned)(rec.prefix + j); }   // TSample_impRank: ranked below
    }
  }
}
// TSample_impRank::prepare (:137-146): the i-th largest error gets 1 / sqrt(sqrt(i + 1)) (double square roots of an integer,
// rounded to float), transitions whose error is not positive get 1
__global__ __launch_bounds__(256) void per_rank_kernel(PerArgs a, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float P = a.keySorted[i] > 0 ? (float)(1 / sqrt(sqrt((double)(i + 1)))) : 1.f;
  a.prob[a.idxSorted[i]] = P;
}

// discrete_distribution::param_type::_M_initialize: sum = accumulate(p, 0.0); p /= sum; cp = partial_sum(p); cp.back() = 1.
// One wavefront: its 64 lanes move blocks of 4096 values between global memory and LDS (coalesced), lane 0 walks a block
// sequentially, sixteen LDS reads ahead of the dependent additions: 23 ms per step on a million transitions -- the chain of
// two million dependent fp64 additions of a lone wavefront (11 ns each; overlapping the LDS reads with the additions changes
// nothing; a first version passed the values lane to lane through v_readlane: 29 ms).  The CPU does the same chain in ~4 ms.
constexpr int PER_BLK = 4096;
__global__ __launch_bounds__(64) void per_scan_seq_kernel(const float* __restrict__ prob, double* __restrict__ cp, long long n) {
  __shared__ double sh[PER_BLK];
  __shared__ double sSum;
  if (n < 2) return;                                   // (the distribution then always returns 0: nothing to build)
  const int lane = threadIdx.x;
  auto sync = [] { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };      // one wavefront: LDS traffic settled
  double s = 0.0;
  for (long long base = 0; base < n; base += PER_BLK) {
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) sh[j] = base + j < n ? (double)prob[base + j] : 0.0;    // (+ 0.0 behind the end: exact)
    sync();
    if (lane == 0) {
      for (int j = 0; j < PER_BLK; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sh[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
    }
    sync();
  }
  if (lane == 0) sSum = s;
  sync();
  s = sSum;
  double acc = 0.0;
  for (long long base = 0; base < n; base += PER_BLK) {
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) sh[j] = base + j < n ? (double)prob[base + j] / s : 0.0;
    sync();
    if (lane == 0) {
      for (int j = 0; j < PER_BLK; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sh[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc += v[u]; v[u] = acc; }
#pragma unroll
        for (int u = 0; u < 16; ++u) sh[j + u] = v[u];
      }
    }
    sync();
#pragma unroll 8
    for (int j = lane; j < PER_BLK; j += 64) { const long long e = base + j; if (e < n) cp[e] = e == n - 1 ? 1.0 : sh[j]; }
    sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same two chains WITHOUT walking them element by element, bit for bit (per_scan_kernel, round 4).
// A chain is acc_i = fl(acc_{i-1} + x_i) in round-to-nearest-even fp64 with x_i > 0.  While acc stays inside one binade
// [2^e, 2^(e+1)) it is an integer multiple M of u = 2^(e-52), and fl(M u + x) = (M + k + r) u with k = floor(x / u) and r = 1 when
// the remainder of x / u is above one half, 0 when below: an INTEGER prefix sum, order-free.  What does not fit that form is done
// as one real fp64 addition by the lane owning the element: the addition that carries acc into the next binade (u doubles; the scan
// finds the first element whose integer reaches 2^53) and an exact tie (remainder == 1/2: the rounding looks at M's parity; about
// one element in 2^20 for quotients, none for sums of floats whose last bit lies above u).  One workgroup of 1024 threads takes the
// table in chunks of 16 elements per thread: saturating int64 scan (thread, wavefront, workgroup), first binade crossing / tie
// by LDS atomics, elements in front of it final, the element itself added in fp64, the rest of the chunk scanned again with the
// new acc.  The first elements (a dozen binades within the first few thousand) are walked by lane 0 as before.
// 1M transitions: both chains in 2.4 ms against 23.4 ms of the walk (per_scan_seq_kernel: SMARTIES_HIP_PER_SEQ=1,
// kept for the test that compares the two tables bit for bit).
// ---------------------------------------------------------------------------------------------------------------
constexpr int PS_NT = 1024, PS_PER = 16, PS_CH = PS_NT * PS_PER, PS_HEAD = 4096;
constexpr long long PS_TOP = 1ll << 53, PS_SAT = 1ll << 54;
__device__ __forceinline__ long long psAdd(long long a, long long b) { const long long s = a + b; return s < PS_SAT ? s : PS_SAT; }      // (associative on non-negative numbers)
template <bool DIV, bool WRITE>
__device__ __forceinline__ double perChain(const float* __restrict__ prob, double S, double* __restrict__ cp, long long n, long long* sWave, double* sAcc, int* sStop, double* sHead,
                                           const long long* __restrict__ cT = nullptr, const int* __restrict__ cF = nullptr, double* __restrict__ accStart = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto elem = [&](long long j) -> double { const double p = (double)prob[j]; return DIV ? p / S : p; };
  const long long head = n < PS_HEAD ? n : PS_HEAD;
  for (int j = tid; j < PS_HEAD; j += PS_NT) sHead[j] = j < head ? elem(j) : 0.0;      // (+ 0.0 behind the end: exact)
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    for (int j = 0; j < PS_HEAD; j += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = sHead[j + q];
#pragma unroll
      for (int q = 0; q < 16; ++q) { a += v[q]; v[q] = a; }
      if (WRITE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sHead[j + q] = v[q];
      }
    }
    *sAcc = a;
  }
  __syncthreads();
  if (WRITE) for (int j = tid; j < head; j += PS_NT) cp[j] = sHead[j];
  double acc = *sAcc;
  for (long long pos = head; pos < n; pos += PS_CH) {
    const int cn = n - pos < PS_CH ? (int)(n - pos) : PS_CH;
    if (cT) {      // grid form: a chunk whose integer total was computed beside (per_chunk_total_kernel) for the binade the accumulator is in, and stays in
      const int ch = (int)((pos - head) / PS_CH), fe = cF[ch];
      const long long bits = __double_as_longlong(acc);
      const int ex = (int)((bits >> 52) & 0x7ff) - 1023;
      bool easy = false;
      if (fe && ex == fe - 4096) {
        const double scale = __longlong_as_double((long long)(1023 + 52 - ex) << 52), u = __longlong_as_double((long long)(1023 + ex - 52) << 52);
        const long long M1 = (long long)(acc * scale) + cT[ch];
        if (M1 < PS_TOP) { easy = true; if (accStart && tid == 0) accStart[ch] = acc; acc = (double)M1 * u; }
      }
      if (easy) continue;                                                 // (uniform: every thread holds the same acc)
      if (accStart && tid == 0) accStart[ch] = -1.0;                      // walked here, written here
    }
    double x[PS_PER];
#pragma unroll
    for (int i = 0; i < PS_PER; ++i) { const int g = tid * PS_PER + i; x[i] = g < cn ? elem(pos + g) : 0.0; }
    int done = 0;
    while (done < cn) {
      const long long bits = __double_as_longlong(acc);
      const int ex = (int)((bits >> 52) & 0x7ff) - 1023;
      const double scale = __longlong_as_double((long long)(1023 + 52 - ex) << 52), u = __longlong_as_double((long long)(1023 + ex - 52) << 52);
      const long long M0 = (long long)(acc * scale);                      // in [2^52, 2^53), exact
      long long loc[PS_PER]; long long run = 0; int firstTie = PS_CH;
#pragma unroll
      for (int i = 0; i < PS_PER; ++i) {
        const int g = tid * PS_PER + i;
        long long cc = 0;
        if (g >= done && g < cn) {
          const double y = x[i] * scale;                                  // exact (power of two)
          if (y >= 9007199254740992.0) cc = PS_SAT;
          else { const long long k = (long long)y; const double f = y - (double)k; cc = k + (f > 0.5 ? 1 : 0); if (f == 0.5 && g < firstTie) firstTie = g; }
        }
        run = psAdd(run, cc); loc[i] = run;
      }
      // exclusive scan of the threads' totals: wavefront, then the sixteen wavefront totals
      long long inc = run;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if (lane >= d) inc = psAdd(inc, o); }
      if (tid == 0) { sStop[0] = PS_CH; }
      if (lane == 63) sWave[wave] = inc;
      __syncthreads();
      const long long left = __shfl_up(inc, 1, 64);                     // exclusive value within the wavefront
      long long off = lane ? left : 0;
      for (int w = 0; w < wave; ++w) off = psAdd(off, sWave[w]);
      // first element of [done, cn) that cannot be taken as an integer step: a binade crossing or an exact tie
      int stop = firstTie;
#pragma unroll
      for (int i = PS_PER - 1; i >= 0; --i) { const int g = tid * PS_PER + i; if (g >= done && g < cn && M0 + psAdd(off, loc[i]) >= PS_TOP) stop = g < stop ? g : stop; }
      if (stop < PS_CH) atomicMin(sStop, stop);
      __syncthreads();
      stop = sStop[0] < cn ? sStop[0] : cn;
      if (WRITE) {
#pragma unroll
        for (int i = 0; i < PS_PER; ++i) { const int g = tid * PS_PER + i; if (g >= done && g < stop) cp[pos + g] = (double)(M0 + off + loc[i]) * u; }
      }
      if (stop < cn) {
        if (stop / PS_PER == tid) {                                       // the owner of element `stop`: one real fp64 addition
          const int i = stop - tid * PS_PER;
          long long before = off;
#pragma unroll
          for (int q = 0; q < PS_PER; ++q) if (q == i - 1) before = off + loc[q];
          double xs = 0.0;
#pragma unroll
          for (int q = 0; q < PS_PER; ++q) if (q == i) xs = x[q];
          const double a = (double)(M0 + before) * u + xs;
          if (WRITE) cp[pos + stop] = a;
          *sAcc = a;
        }
        done = stop + 1;
      } else {
        if (tid == (cn - 1) / PS_PER) {
          const int i = cn - 1 - tid * PS_PER;
          long long last = 0;
#pragma unroll
          for (int q = 0; q < PS_PER; ++q) if (q == i) last = off + loc[q];
          *sAcc = (double)(M0 + last) * u;
        }
        done = cn;
      }
      __syncthreads();
      acc = *sAcc;
    }
  }
  return acc;
}
// ---- the grid form for long tables: the chunks' integer totals are computed by one workgroup per chunk beside each other, for the
// binade a parallel (approximate) prefix sum PREDICTS for the chunk; the serial pass then steps over a chunk with one addition of
// integers after CHECKING the prediction exactly (accumulator in that binade before and after the chunk, no tie inside) and walks
// the others -- those with a binade crossing or a tie, and any the prediction got wrong -- as above; the cumulative values of the
// stepped-over chunks are written by a grid again.  Scratch (doubles): [0] sum of the table, [1 .. nB] approximate sums of the head
// and of the chunks, [1 + nB ..] the accumulator at the chunks' starts; behind them the int64 totals and the int flags.
template <bool DIV> __device__ __forceinline__ double psElemAt(const float* __restrict__ prob, double S, long long j) { const double p = (double)prob[j]; return DIV ? p / S : p; }
__device__ __forceinline__ double psBlockSumD(double v, double* sRed) {      // (any order: only the prediction uses it)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < PS_NT / 64; ++w) s += sRed[w];
  return s;
}
template <bool DIV>
__global__ __launch_bounds__(PS_NT) void per_chunk_approx_kernel(const float* __restrict__ prob, const double* __restrict__ scr, double* __restrict__ approx, long long n) {
  __shared__ double sRed[PS_NT / 64];
  const long long head = n < PS_HEAD ? n : PS_HEAD;
  const double S = DIV ? scr[0] : 1.0;
  const long long b0 = blockIdx.x == 0 ? 0 : head + (long long)(blockIdx.x - 1) * PS_CH;
  const long long b1 = blockIdx.x == 0 ? head : (b0 + PS_CH < n ? b0 + PS_CH : n);
  double v = 0.0;
  for (long long j = b0 + threadIdx.x; j < b1; j += PS_NT) v += psElemAt<DIV>(prob, S, j);
  v = psBlockSumD(v, sRed);
  if (threadIdx.x == 0) approx[blockIdx.x] = v;
}
template <bool DIV>
__global__ __launch_bounds__(PS_NT) void per_chunk_total_kernel(const float* __restrict__ prob, const double* __restrict__ scr, const double* __restrict__ approx,
                                                               long long* __restrict__ cT, int* __restrict__ cF, long long n) {
  __shared__ double sRed[PS_NT / 64];
  __shared__ long long sTot[PS_NT / 64];
  __shared__ int sTie;
  const int ch = blockIdx.x, tid = threadIdx.x;
  const long long head = n < PS_HEAD ? n : PS_HEAD;
  const double S = DIV ? scr[0] : 1.0;
  double v = 0.0;
  for (int b = tid; b <= ch; b += PS_NT) v += approx[b];                  // head + the chunks in front of this one
  if (tid == 0) sTie = 0;
  const double s0 = psBlockSumD(v, sRed), s1 = s0 + approx[ch + 1];
  const double lo = s0 * (1.0 - 1e-9), hi = s1 * (1.0 + 1e-9);
  const int e0 = (int)((__double_as_longlong(lo) >> 52) & 0x7ff) - 1023, e1 = (int)((__double_as_longlong(hi) >> 52) & 0x7ff) - 1023;
  if (e0 != e1 || !(lo > 0.0)) { if (tid == 0) { cF[ch] = 0; cT[ch] = 0; } return; }      // (uniform)
  const double scale = __longlong_as_double((long long)(1023 + 52 - e0) << 52);
  const long long pos = head + (long long)ch * PS_CH;
  const int cn = n - pos < PS_CH ? (int)(n - pos) : PS_CH;
  long long run = 0; bool tie = false;
#pragma unroll
  for (int i = 0; i < PS_PER; ++i) {
    const int g = tid * PS_PER + i;
    if (g < cn) {
      const double y = psElemAt<DIV>(prob, S, pos + g) * scale;
      if (y >= 9007199254740992.0) run = PS_SAT;
      else { const long long k = (long long)y; const double f = y - (double)k; run = psAdd(run, k + (f > 0.5 ? 1 : 0)); tie = tie || f == 0.5; }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) run = psAdd(run, __shfl_xor(run, d, 64));
  if ((tid & 63) == 0) sTot[tid >> 6] = run;
  if (tie) sTie = 1;
  __syncthreads();
  if (tid == 0) {
    long long t = 0;
    for (int w = 0; w < PS_NT / 64; ++w) t = psAdd(t, sTot[w]);
    cT[ch] = t; cF[ch] = (sTie || t >= PS_TOP) ? 0 : e0 + 4096;
  }
}
// the serial pass (one workgroup); WRITE: the cumulative values of the head and of the chunks it walks, else the table's sum to scr[0]
template <bool DIV, bool WRITE>
__global__ __launch_bounds__(PS_NT) void per_chain_grid_kernel(const float* __restrict__ prob, double* __restrict__ scr, double* __restrict__ cp, long long n,
                                                              const long long* __restrict__ cT, const int* __restrict__ cF, double* __restrict__ accStart) {
  __shared__ long long sWave[PS_NT / 64];
  __shared__ double sAcc;
  __shared__ int sStop[1];
  __shared__ double sHead[PS_HEAD];
  const double S = perChain<DIV, WRITE>(prob, DIV ? scr[0] : 1.0, cp, n, sWave, &sAcc, sStop, sHead, cT, cF, WRITE ? accStart : nullptr);
  __syncthreads();
  if (threadIdx.x == 0) { if (WRITE) cp[n - 1] = 1.0; else scr[0] = S; }
}
// the cumulative values of the chunks the serial pass stepped over
template <bool DIV>
__global__ __launch_bounds__(PS_NT) void per_chunk_write_kernel(const float* __restrict__ prob, const double* __restrict__ scr, const double* __restrict__ accStart,
                                                               double* __restrict__ cp, long long n) {
  __shared__ long long sWave[PS_NT / 64];
  const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double acc = accStart[ch];
  if (!(acc > 0.0)) return;                                               // walked (and written) by the serial pass
  const long long head = n < PS_HEAD ? n : PS_HEAD;
  const double S = DIV ? scr[0] : 1.0;
  const int ex = (int)((__double_as_longlong(acc) >> 52) & 0x7ff) - 1023;
  const double scale = __longlong_as_double((long long)(1023 + 52 - ex) << 52), u = __longlong_as_double((long long)(1023 + ex - 52) << 52);
  const long long M0 = (long long)(acc * scale);
  const long long pos = head + (long long)ch * PS_CH;
  const int cn = n - pos < PS_CH ? (int)(n - pos) : PS_CH;
  long long loc[PS_PER]; long long run = 0;
#pragma unroll
  for (int i = 0; i < PS_PER; ++i) {
    const int g = tid * PS_PER + i;
    if (g < cn) { const double y = psElemAt<DIV>(prob, S, pos + g) * scale; const long long k = (long long)y; const double f = y - (double)k; run += k + (f > 0.5 ? 1 : 0); }
    loc[i] = run;
  }
  long long inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
  if (lane == 63) sWave[wave] = inc;
  __syncthreads();
  const long long left = __shfl_up(inc, 1, 64);
  long long off = lane ? left : 0;
  for (int w = 0; w < wave; ++w) off += sWave[w];
#pragma unroll
  for (int i = 0; i < PS_PER; ++i) { const int g = tid * PS_PER + i; if (g < cn) cp[pos + g] = pos + g == n - 1 ? 1.0 : (double)(M0 + off + loc[i]) * u; }
}
size_t per_scan_scratch_bytes(long long n) {
  const long long nCh = n > PS_HEAD ? (n - PS_HEAD + PS_CH - 1) / PS_CH : 0;
  return (size_t)(1 + (nCh + 1) + nCh) * 8 + (size_t)nCh * 8 + (size_t)nCh * 4 + 64;
}
constexpr long long PS_GRID_MIN = PS_HEAD + 8 * PS_CH;      // shorter tables: the one-workgroup form
static hipError_t launchPerScanGrid(const float* prob, double* cp, long long n, void* scratch, hipStream_t s) {
  const long long nCh = (n - PS_HEAD + PS_CH - 1) / PS_CH;
  double* scr = reinterpret_cast<double*>(scratch);
  double* approx = scr + 1; double* accStart = approx + (nCh + 1);
  long long* cT = reinterpret_cast<long long*>(accStart + nCh); int* cF = reinterpret_cast<int*>(cT + nCh);
  const unsigned g = (unsigned)nCh;
  hipLaunchKernelGGL(per_chunk_approx_kernel<false>, dim3(g + 1), dim3(PS_NT), 0, s, prob, scr, approx, n);
  hipLaunchKernelGGL(per_chunk_total_kernel<false>, dim3(g), dim3(PS_NT), 0, s, prob, scr, approx, cT, cF, n);
  hipLaunchKernelGGL((per_chain_grid_kernel<false, false>), dim3(1), dim3(PS_NT), 0, s, prob, scr, cp, n, cT, cF, accStart);
  hipLaunchKernelGGL(per_chunk_approx_kernel<true>, dim3(g + 1), dim3(PS_NT), 0, s, prob, scr, approx, n);
  hipLaunchKernelGGL(per_chunk_total_kernel<true>, dim3(g), dim3(PS_NT), 0, s, prob, scr, approx, cT, cF, n);
  hipLaunchKernelGGL((per_chain_grid_kernel<true, true>), dim3(1), dim3(PS_NT), 0, s, prob, scr, cp, n, cT, cF, accStart);
  hipLaunchKernelGGL(per_chunk_write_kernel<true>, dim3(g), dim3(PS_NT), 0, s, prob, scr, accStart, cp, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(PS_NT) void per_scan_kernel(const float* __restrict__ prob, double* __restrict__ cp, long long n) {
  __shared__ long long sWave[PS_NT / 64];
  __shared__ double sAcc;
  __shared__ int sStop[1];
  __shared__ double sHead[PS_HEAD];
  if (n < 2) return;                                   // (the distribution then always returns 0: nothing to build)
  const double S = perChain<false, false>(prob, 1.0, nullptr, n, sWave, &sAcc, sStop, sHead);
  __syncthreads();
  perChain<true, true>(prob, S, cp, n, sWave, &sAcc, sStop, sHead);
  __syncthreads();
  if (threadIdx.x == 0) cp[n - 1] = 1.0;
}

size_t per_sort_temp_bytes(long long n) {
  size_t bytes = 0;
  hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, (int)n);
  return bytes;
}
hipError_t launch_per_prepare(const PerArgs& a, long long nTransitions, hipStream_t s) {
  int nb = (a.nEpisodes + 3) / 4; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(per_probs_kernel, dim3(nb), dim3(256), 0, s, a);
  long long n = a.algo == HL_SAMPLE_PERSEQ ? a.nEpisodes : nTransitions;
  if (a.algo == HL_SAMPLE_PERRANK && n > 0) {
    size_t bytes = a.tempBytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(a.temp, bytes, a.key, a.keySorted, a.idx, a.idxSorted, (int)n, 0, 32, s);   // stable
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(per_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n);
  }
  static const bool seq = [] { const char* e = getenv("SMARTIES_HIP_PER_SEQ"); return e && atoi(e) != 0; }();
  if (seq) hipLaunchKernelGGL(per_scan_seq_kernel, dim3(1), dim3(64), 0, s, a.prob, a.cp, n);
  else if (n >= PS_GRID_MIN && a.scan && a.scanBytes >= per_scan_scratch_bytes(n)) return launchPerScanGrid(a.prob, a.cp, n, a.scan, s);
  else hipLaunchKernelGGL(per_scan_kernel, dim3(1), dim3(PS_NT), 0, s, a.prob, a.cp, n);
  return hipGetLastError();
}

// the table of an arbitrary probability array (tests / tools): which = 0 the scan above, 1 the sequential walk
hipError_t launch_per_scan(const float* prob, double* cp, long long n, int which, void* scratch, hipStream_t s) {
  if (which == 1) hipLaunchKernelGGL(per_scan_seq_kernel, dim3(1), dim3(64), 0, s, prob, cp, n);
  else if (which == 0 && scratch && n >= PS_GRID_MIN) return launchPerScanGrid(prob, cp, n, scratch, s);
  else hipLaunchKernelGGL(per_scan_kernel, dim3(1), dim3(PS_NT), 0, s, prob, cp, n);      // (2: the one-workgroup form whatever the length)
  return hipGetLastError();
}

}  // namespace hl
