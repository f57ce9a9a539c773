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
// one of two pointers of a problem record by a per-lane condition, selected on the VALUES (two v_cndmask): written as `c ? P.a : P.b`
// the compiler loads through a selected ADDRESS of the record's fields, which puts the record's pointers on the stack -- the 36 - 40
// bytes of scratch per lane dw_table_kernel, fused_wide_kernel and panel_head_kernel carried until round 5
__device__ __forceinline__ const float* pickPtr(bool first, const float* a, const float* b) {
  unsigned long long ua = reinterpret_cast<unsigned long long>(a), ub = reinterpret_cast<unsigned long long>(b);
  asm volatile("" : "+s"(ua), "+s"(ub));      // (the two values exist in registers before the select)
  return reinterpret_cast<const float*>(first ? ua : ub);
}
__device__ __forceinline__ float* pickPtrW(bool first, float* a, float* b) { return const_cast<float*>(pickPtr(first, a, b)); }
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
// which of its two arguments actDiff reads: the layer's output for these, its pre-activation for the others (never both)
__host__ __device__ __forceinline__ bool actDiffFromOutput(int f) { return f == HL_FUNC_TANH || f == HL_FUNC_SIGM || f == HL_FUNC_EXP; }
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
// ---------------------------------------------------------------------------------------------
// ReplayStats::nFarPolicySteps, bit for bit.  MemoryProcessing::updateTrainingStatistics (MemoryProcessing.cpp:205-227)
// adds `Nsteps * fracFarPolSteps` -- a float, fractional in general (fractions are formed over ndata, weighted with nsteps) --
// to a Uint with `+=`: the running count is converted to float, product and sum are formed in float, the result is truncated,
// episode by episode in storage order.  Neither a sum of per-episode integers nor any tree reproduces that (k / N * N lands
// half an ulp below k, the add rounds it back up or not depending on the running count), so the loop is kept: thread t walks
// its contiguous segment of episodes from a start value, the segment sums are scanned, and the walk is repeated with the
// scanned starts until they stop changing -- the fixed point IS the sequential result; the starts of the previous step are
// the first guess, so two passes (one to compute, one to confirm) are the rule.
//   Multiply and add are ONE fused operation here: the build of the reference the fixtures come from (oracle/Makefile:
//   -march=x86-64-v3, gcc's default -ffp-contract=fast) compiles the statement to vfmadd, and so does the oracle's; a build
//   of the reference without FMA rounds the product first and differs where the sum falls on a tie (seen: 8 + 5.9999995).
// Counts below 2^30 with non-negative terms take hardware conversions (three dependent instructions per episode); anything
// else -- the first steps with clipImpWeight < 1, where fractions go negative and the reference's float -> Uint conversion is
// what x86 makes of it -- the emulated 64-bit conversions.
// Fractions and lengths live in DevReplay::farP / farN in the walk's layout (element i of thread t's segment at [i * 256 + t],
// 256 segments of per = ceil(nEp / 256) consecutive positions): rebuilt when the table or all fractions change
// (far_build_kernel), the fractions patched by the bookkeeping pass for the episodes of the minibatch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long cvtFloatToUintX86(float x) {      // gcc's float -> uint64 on x86-64 (cvttss2si based)
  auto cvtt = [](float v) -> long long { return (v >= -9223372036854775808.f && v < 9223372036854775808.f) ? (long long)v : (long long)0x8000000000000000ull; };
  if (x < 9223372036854775808.f) return (unsigned long long)cvtt(x);           // (NaN compares false: second branch, as the compiled code)
  return (unsigned long long)cvtt(x - 9223372036854775808.f) ^ 0x8000000000000000ull;
}
// the iteration around a walker `walk(n0) -> n` over this thread's segment (sScan: four 64-bit words of LDS)
template <class WALK> __device__ __forceinline__ unsigned long long farFixedPoint(WALK walk, unsigned long long* gStart, unsigned long long* sScan) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  unsigned long long n0 = gStart[t], total = 0;
  for (int iter = 0; iter < 258; ++iter) {             // (segment s is final after s + 1 rounds whatever the guesses were)
    const unsigned long long d = walk(n0) - n0;
    // exclusive scan of the segment sums: wave scan, then the four wave totals
    unsigned long long inc = d;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    __syncthreads();                                   // (sScan of the previous round has been read)
    if (lane == 63) sScan[wv] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < wv; ++w) base += sScan[w];
    const unsigned long long start = base + inc - d;
    total = sScan[0] + sScan[1] + sScan[2] + sScan[3];
    const int changed = __syncthreads_or(start != n0 ? 1 : 0);
    n0 = start;
    if (!changed) break;
  }
  gStart[t] = n0;
  return total;
}
// the reference's loop over one segment: fractions and lengths from memory (element i of thread t at [i * 256 + t]; COHERENT:
// fractions written by global stores of this kernel) ...
template <bool COHERENT> __device__ __forceinline__ unsigned long long farWalkMem(const float* F, const float* L, int cnt, unsigned long long n0) {
  auto ld = [&](int i) -> float {
    const float* q = F + (size_t)i * 256 + threadIdx.x;
    return COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
  };
  if (n0 < 0x40000000ull) {
    unsigned m = (unsigned)n0; bool ok = true;
    for (int i = 0; i < cnt && ok; ++i) {
      const float x = __builtin_fmaf(L[(size_t)i * 256 + threadIdx.x], ld(i), (float)m);
      ok = x >= 0.f && x < 1073741824.f;
      m = ok ? (unsigned)x : 0u;
    }
    if (ok) return m;
  }
  unsigned long long n = n0;
  for (int i = 0; i < cnt; ++i) n = cvtFloatToUintX86(__builtin_fmaf(L[(size_t)i * 256 + threadIdx.x], ld(i), (float)n));
  return n;
}
// ... or from registers (FAR_REGS episodes per thread at most; unused slots hold length 0): the bookkeeping rider's walk.  While
// the count stays below 2^24 it is carried as a float -- (float)(Uint)x == truncf(x) for 0 <= x < 2^24, and a sum in (-1, 0)
// truncates to zero either way --, which leaves two dependent instructions per episode (v_fma_f32, v_trunc_f32); the smallest
// sum seen and the final count tell whether that held.  A thread's segment is cut into FAR_SUB pieces walked side by side from
// their own start values (four independent dependency chains: the walk is latency bound, one wavefront per SIMD), so the
// fixed point runs over 4 x 256 pieces; DevReplay::farStart keeps the start of piece c of thread t at [c * 256 + t].
__device__ __forceinline__ bool farWalkRegs(const float (&f)[FAR_REGS], const float (&l)[FAR_REGS], const unsigned (&n0)[FAR_SUB], unsigned (&n)[FAR_SUB]) {
  float nf[FAR_SUB], lo = 0.f;
#pragma unroll
  for (int c = 0; c < FAR_SUB; ++c) nf[c] = (float)n0[c];
#pragma unroll
  for (int i = 0; i < FAR_Q; ++i) {
#pragma unroll
    for (int c = 0; c < FAR_SUB; ++c) {
      const float x = __builtin_fmaf(l[c * FAR_Q + i], f[c * FAR_Q + i], nf[c]);
      lo = fminf(lo, x);
      nf[c] = __builtin_truncf(x);
    }
  }
  bool ok = lo > -1.f;
#pragma unroll
  for (int c = 0; c < FAR_SUB; ++c) {
    ok = ok && n0[c] < 16777216u && nf[c] < 16777216.f;       // (a NaN fraction fails the last comparison)
    n[c] = (unsigned)fminf(fmaxf(nf[c], 0.f), 16777216.f);
  }
  return ok;
}
// inclusive scan over the 64 lanes of a wavefront in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16, then lane 15 of
// rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3)
__device__ __forceinline__ unsigned waveScanIncl(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}
// the fixed point around the register walk, counts in 32 bits; g0: the pieces' starts of the last pass (fetched early).
// false: left the range of the float recurrence (any thread, any round) -- the caller runs farCountMem instead.  One barrier
// per round: the wavefronts exchange their totals together with "an increment differs from the round before" / "out of range"
// in one LDS word each (sX: eight words, two rounds alternate); a round that reproduces all increments ends the iteration.
__device__ __forceinline__ bool farCountRegs(const float (&f)[FAR_REGS], const float (&l)[FAR_REGS], const unsigned long long (&g0)[FAR_SUB],
                                             unsigned long long* gStart, unsigned* sX, unsigned long long* total) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  unsigned n0[FAR_SUB], n[FAR_SUB], dPrev[FAR_SUB], tot = 0; int iterDone = 0; (void)iterDone;
#pragma unroll
  for (int c = 0; c < FAR_SUB; ++c) { n0[c] = g0[c] < 16777216ull ? (unsigned)g0[c] : 0u; dPrev[c] = 0u; }
  for (int iter = 0; iter < 1030; ++iter) {
    const bool ok = farWalkRegs(f, l, n0, n);
    unsigned d = 0; bool same = iter > 0;
#pragma unroll
    for (int c = 0; c < FAR_SUB; ++c) { const unsigned dc = n[c] - n0[c]; same = same && dc == dPrev[c]; dPrev[c] = dc; d += dc; }
    const unsigned inc = waveScanIncl(d);
    const unsigned fl = (__ballot(!same) != 0ull ? 0x40000000u : 0u) | (__ballot(!ok) != 0ull ? 0x80000000u : 0u);
    unsigned* x = sX + (iter & 1) * 4;
    if (lane == 63) x[wv] = (inc & 0x3fffffffu) | fl;
    __syncthreads();
    const unsigned w0 = x[0], w1 = x[1], w2 = x[2], w3 = x[3], any = w0 | w1 | w2 | w3;
    if (any & 0x80000000u) return false;
    tot = (w0 & 0x3fffffffu) + (w1 & 0x3fffffffu) + (w2 & 0x3fffffffu) + (w3 & 0x3fffffffu);
    if (!(any & 0x40000000u)) { iterDone = iter + 1; break; }         // the starts of this round came from exactly these increments
    unsigned start = inc - d + (wv > 0 ? (w0 & 0x3fffffffu) : 0u) + (wv > 1 ? (w1 & 0x3fffffffu) : 0u) + (wv > 2 ? (w2 & 0x3fffffffu) : 0u);
#pragma unroll
    for (int c = 0; c < FAR_SUB; ++c) { n0[c] = start; start += dPrev[c]; }
  }
#pragma unroll
  for (int c = 0; c < FAR_SUB; ++c) gStart[c * 256 + t] = n0[c];
  *total = tot;
#ifdef HL_TAIL_STAMPS
  *total |= (unsigned long long)iterDone << 48;
#endif
  return true;
}
// the whole count over terms in memory
template <bool COHERENT> __device__ __forceinline__ unsigned long long farCountMem(const float* F, const float* L, int cnt, unsigned long long* gStart, unsigned long long* sScan) {
  return farFixedPoint([F, L, cnt](unsigned long long n0) { return farWalkMem<COHERENT>(F, L, cnt, n0); }, gStart, sScan);
}
__device__ __forceinline__ int farSegment(int nEp, int per) { return min(per, max(0, nEp - (int)threadIdx.x * per)); }

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

__device__ __forceinline__ float resOut(float y, float in, float w, float b) { return y + fmaf(in, w, b); }
// Cross-lane traffic inside a 16-lane row (one sample) goes through DPP row rotations -- one VALU
// instruction each -- instead of ds_bpermute round trips through the LDS crossbar.
// rowRor<N>: lane i of a row receives the value of lane (i - N) mod 16 of the same row.
template <int N> __device__ __forceinline__ int rowRorI(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false); }
template <int N> __device__ __forceinline__ float rowRorF(float v) { return __int_as_float(rowRorI<N>(__float_as_int(v))); }
template <int N> __device__ __forceinline__ double rowRorD(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)rowRorI<N>((int)(unsigned)b), hi = (unsigned)rowRorI<N>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the 16 lanes of one sample, every lane gets it.  Same pairing as an xor butterfly
// (8, 4, 2, 1: the partial sums are periodic, so rotating or exchanging gives the same operands).
__device__ __forceinline__ double sum16(double v) {
  v += rowRorD<8>(v); v += rowRorD<4>(v); v += rowRorD<2>(v); v += rowRorD<1>(v);
  return v;
}
// value of lane 0 of the row in every lane (exact: the other lanes contribute +0)
__device__ __forceinline__ float bcast0F(float v, int en) {
  float x = en == 0 ? v : 0.f;
  x += rowRorF<8>(x); x += rowRorF<4>(x); x += rowRorF<2>(x); x += rowRorF<1>(x);
  return x;
}

// activation evaluated with the function known at compile time; dispatchFunc() branches ONCE on the
// (uniform) function id and runs the whole epilogue branch-free
// n / d for d in [1, 2^60): reciprocal + Newton step + two residual corrections -- the IEEE division
// expansion without its range scaling (v_div_scale / v_div_fmas serialise on VCC, so sixteen of
// them per thread cannot overlap; these can)
__device__ __forceinline__ float divNoScale(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  float q = n * r;
  q = fmaf(fmaf(-d, q, n), r, q);
  q = fmaf(fmaf(-d, q, n), r, q);
  return q;
}
template <int F> __device__ __forceinline__ float actEvalT(float in) {
  if constexpr (F == HL_FUNC_SOFTSIGN) return divNoScale(in, 1 + fabsf(in));
  else return actEval(F, in);
}
template <int F> __device__ __forceinline__ float actDiffT(float in, float out) {
  if constexpr (F == HL_FUNC_SOFTSIGN) { const float d = 1 + fabsf(in); return divNoScale(1.0f, d * d); }
  else return actDiff(F, in, out);
}
template <int F> struct FuncTag { static constexpr int value = F; };
// CF >= 0: the activation is a template parameter of the kernel (no other code is generated)
template <int CF, class Body> __device__ __forceinline__ void dispatchFunc(int func, Body body) {
  if constexpr (CF >= 0) body(FuncTag<CF>{});
  else if (func == HL_FUNC_SOFTSIGN) body(FuncTag<HL_FUNC_SOFTSIGN>{});
  else if (func == HL_FUNC_TANH) body(FuncTag<HL_FUNC_TANH>{});
  else if (func == HL_FUNC_RELU) body(FuncTag<HL_FUNC_RELU>{});
  else if (func == HL_FUNC_LRELU) body(FuncTag<HL_FUNC_LRELU>{});
  else if (func == HL_FUNC_SIGM) body(FuncTag<HL_FUNC_SIGM>{});
  else if (func == HL_FUNC_HARDSIGN) body(FuncTag<HL_FUNC_HARDSIGN>{});
  else if (func == HL_FUNC_SOFTPLUS) body(FuncTag<HL_FUNC_SOFTPLUS>{});
  else if (func == HL_FUNC_EXPPLUS) body(FuncTag<HL_FUNC_EXPPLUS>{});
  else if (func == HL_FUNC_EXP) body(FuncTag<HL_FUNC_EXP>{});
  else body(FuncTag<HL_FUNC_LINEAR>{});
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
