/*
 * oracle/port/vracer_port.cpp -- CPU restatement of the smarties V-RACER /
 * ReF-ER learner update.
 *
 * TEST INFRASTRUCTURE ONLY (the oracle).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the
 * checker; the product path (smarties_amd/csrc, libsmarties_hip.so) never does.
 *
 * Parity status: PINNED -- every function below is checked against golden
 * fixtures produced by the compiled reference itself (oracle/_ref/ref_driver,
 * generator script tests/golden/make_golden.sh; tests/test_oracle_golden.py).
 *
 * Written from the reference's behaviour, not from its text; each block cites
 * the reference lines it restates (paths relative to
 * /root/reference/source/smarties/).  Single-threaded: it restates the
 * reference run with nThreads = 1, which fixes every summation order.
 */
#include "vracer_port.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <tuple>
#include <limits>
#include <vector>

namespace {

using Real = double;   // Settings/Definitions.h:27
using nnReal = float;  // Settings/Definitions.h:46 (-DSINGLE_PREC)
using Fval = float;    // Settings/Definitions.h:50

constexpr nnReal nnEPS = FLT_EPSILON;  // Settings/Bund.h:113

inline int64_t roundUp8(int64_t n) { return (n + 7) / 8 * 8; }  // Utils/FunctionUtilities.h:74-83

// ---------------------------------------------------------------------------
// std::mt19937 restated (32-bit Mersenne twister, same state layout as
// libstdc++: 624 words + position; position 624 == "regenerate on next draw").
// ---------------------------------------------------------------------------
struct MT19937 {
  uint32_t x[624];
  uint32_t p;
  void seed(uint32_t s) {
    x[0] = s;
    for (uint32_t i = 1; i < 624; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + i;
    p = 624;
  }
  void twist() {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    for (int k = 0; k < 227; ++k) {
      const uint32_t y = (x[k] & UP) | (x[k + 1] & LO);
      x[k] = x[k + 397] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    }
    for (int k = 227; k < 623; ++k) {
      const uint32_t y = (x[k] & UP) | (x[k + 1] & LO);
      x[k] = x[k - 227] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    }
    const uint32_t y = (x[623] & UP) | (x[0] & LO);
    x[623] = x[396] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    p = 0;
  }
  uint32_t next() {
    if (p >= 624) twist();
    uint32_t z = x[p++];
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
  }
};

// std::uniform_int_distribution<size_t>(0, N-1) over mt19937 for N < 2^32 in
// libstdc++ >= 10: Lemire's nearly-divisionless method on 32-bit words
// (call site ReplayMemory/Sampling.cpp:84-88).
inline uint64_t uniformIndex(MT19937& g, uint64_t N) {
  const uint32_t range = (uint32_t)N;
  uint64_t product = (uint64_t)g.next() * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)g.next() * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return product >> 32;
}

// std::uniform_real_distribution<float>(a,b) over mt19937 in libstdc++:
// generate_canonical<float,24> consumes one 32-bit word (call site
// Network/Layers/Layer_Base.h:120-133).
inline float uniformFloat(MT19937& g, float a, float b) {
  float ret = (float)g.next() / 4294967296.0f;
  if (ret >= 1.0f) ret = std::nextafter(1.0f, 0.0f);
  return std::fmaf(ret, b - a, a);   // contracted to an FMA in the reference build (GCC, -march with FMA)
}

// ---------------------------------------------------------------------------
// activation functions (Network/Layers/Functions.h)
// ---------------------------------------------------------------------------
// Utilities::safeExp (Utils/FunctionUtilities.h:51-54) with SMARTIES_EXP_CUT = 8 of the single-precision build (Definitions.h:43)
inline nnReal nnSafeExp(nnReal v) { return std::exp(std::min((nnReal)8, std::max(-(nnReal)8, v))); }
inline nnReal fEval(int f, nnReal in) {
  switch (f) {
    case HL_FUNC_LINEAR: return in;                                   // :40-82
    case HL_FUNC_TANH:                                                // :104-113
      if (in > 0) { const nnReal e = std::exp(-2 * in); return (1 - e) / (1 + e); }
      else        { const nnReal e = std::exp( 2 * in); return (e - 1) / (1 + e); }
    case HL_FUNC_SOFTSIGN: return in / (1 + std::fabs(in));           // :328-331
    case HL_FUNC_RELU: return in > 0 ? in : 0;                        // :415-418
    case HL_FUNC_LRELU: return in > 0 ? in : (nnReal)0.1 * in;         // :461-464 (PRELU_FAC = 0.1, :16-18)
    case HL_FUNC_SIGM:                                                // :158-165
      if (in > 0) return 1 / (1 + nnSafeExp(-in));
      else { const nnReal ex = nnSafeExp(in); return ex / (1 + ex); }
    case HL_FUNC_HARDSIGN: return in / std::sqrt(1 + in * in);         // :220-223
    case HL_FUNC_SOFTPLUS: return (in + std::sqrt(1 + in * in)) / 2;   // :552-555
    case HL_FUNC_EXPPLUS: return std::log(1 + nnSafeExp(in));          // :507-510
    case HL_FUNC_EXP: return nnSafeExp(in);                            // :604-607
  }
  return in;
}
inline nnReal fDiff(int f, nnReal in, nnReal out) {
  switch (f) {
    case HL_FUNC_LINEAR: return 1;
    case HL_FUNC_TANH: return 1 - out * out;                          // :119-122
    case HL_FUNC_SOFTSIGN: { const nnReal d = 1 + std::fabs(in); return 1 / (d * d); }  // :333-337
    case HL_FUNC_RELU: return in > 0 ? 1 : 0;
    case HL_FUNC_LRELU: return in > 0 ? 1 : (nnReal)0.1;                 // :465-468
    case HL_FUNC_SIGM: return out * (1 - out);                          // :179-182
    case HL_FUNC_HARDSIGN: { const nnReal d = std::sqrt(1 + in * in); return 1 / (d * d * d); }   // :225-229
    case HL_FUNC_SOFTPLUS: return (1 + in / std::sqrt(1 + in * in)) / 2;  // :560-563
    case HL_FUNC_EXPPLUS: return 1 / (1 + nnSafeExp(-in));              // :515-518
    case HL_FUNC_EXP: return out;                                       // :614-617
  }
  return 1;
}
inline Real fInitFactor(int f, int inps, int outs) {
  switch (f) {
    case HL_FUNC_LINEAR: return std::sqrt(1. / inps);                 // :43-46
    case HL_FUNC_TANH: return std::sqrt(6. / (inps + outs));          // :94-97
    case HL_FUNC_SOFTSIGN: return std::sqrt(6.0 / (inps + outs));     // :318-321
    case HL_FUNC_RELU: return std::sqrt(2. / inps);                   // :404-407
    case HL_FUNC_LRELU: return std::sqrt(1.0 / inps);                 // :453-460
    case HL_FUNC_SIGM: case HL_FUNC_HARDSIGN: return std::sqrt(6. / (inps + outs));   // :148-156, :210-218
    case HL_FUNC_SOFTPLUS: case HL_FUNC_EXPPLUS: case HL_FUNC_EXP: return std::sqrt(2. / inps);   // :544-551, :496-501, :589-597
  }
  return 1;
}
// Function::inverse (Functions.h: Linear :80, Tanh :114, Sigm :167, HardSign :244, SoftSign :353, Relu :439, LRelu :482,
// ExpPlus :502, SoftPlus :564, Exp :632), in nnReal like the reference: the initial bias of a layer is the pre-image of
// the requested initial output (Layer_Base.h:122-125)
inline nnReal fInverse(int f, nnReal in) {
  switch (f) {
    case HL_FUNC_LINEAR: return in;
    case HL_FUNC_TANH: return std::log((1 + in) / (1 - in)) / 2;
    case HL_FUNC_SIGM: return -std::log(1 / in - 1);
    case HL_FUNC_HARDSIGN: return in / std::sqrt(1 - in * in);
    case HL_FUNC_SOFTSIGN: return in / (1 - std::fabs(in));
    case HL_FUNC_RELU: return in;
    case HL_FUNC_LRELU: return in >= 0 ? in : in / (nnReal)0.1;
    case HL_FUNC_EXPPLUS: return std::log(nnSafeExp(in) - 1);
    case HL_FUNC_SOFTPLUS: return (in * in - (nnReal)0.25) / in;
    case HL_FUNC_EXP: return std::log(in);
  }
  return in;
}
// SoftPlus used by the policy for the stdev ("cheap softplus", Functions.h:552-568)
inline Real spEval(Real in) { return (in + std::sqrt(1 + in * in)) / 2; }
inline Real spDiff(Real in) { return (1 + in / std::sqrt(1 + in * in)) / 2; }
inline Real spInv(Real in) { return (in * in - 0.25) / in; }

// value squashing (Learners/RACER_common.cpp:18-32)
inline Real scaleNet2V(Real x) {
  if (x > 0) return 100 * (x + 51) - 100 * std::sqrt(2601 + 100 * x);
  else       return 100 * (x - 51) + 100 * std::sqrt(2601 - 100 * x);
}
inline Real scaleVdiff(Real x) {
  if (x > 0) return 100 - 5000 / std::sqrt(2601 + 100 * x);
  else       return 100 - 5000 / std::sqrt(2601 - 100 * x);
}
// ReplayMemory/Episode.h:28-33 -- evaluated in Fval
inline bool isFarPolicy(Fval W, Fval C, Fval invC) {
  const bool isOff = W > C || W < invC;
  return C > (Fval)1 && isOff;
}

// ---------------------------------------------------------------------------
// network description (Network/Builder.cpp:48-117, Layers/*.h)
// ---------------------------------------------------------------------------
enum LType { L_INPUT, L_DENSE, L_PARAMRES, L_PARAM, L_LSTM, L_MGU, L_CONV, L_JOIN };
struct Layer {
  LType type; int size = 0, nIn = 0, nOutSimd = 0, func = HL_FUNC_LINEAR;
  hl_conv2d cv{};                     // L_CONV: Conv2DLayer<SoftSign, ...> (Network/Layers/Layer_Conv2D.h:29-232)
  bool bOutput = false, skipInpGrad = false;
  bool rec = false;                   // L_DENSE with a recurrent term: BaseLayer(bRecurrent) for nnType "RNN" (Builder.cpp:76-81, Layer_Base.h:83-93)
  int64_t indW = 0, nW = 0, indB = 0, nB = 0;
  std::vector<Real> biasInit;  // ParamLayer initial values
};

// work memory of one time step (Network/Layers/Activation.h): pre-activations, outputs, errors per layer; an LSTM
// layer uses 4 x nCells entries of each (Layer_LSTM.h:29-48)
struct Act { std::vector<std::vector<nnReal>> X, Y, E; };

inline int actSize(const Layer& l) { return l.type == L_LSTM ? 4 * l.size : (l.type == L_MGU ? 2 * l.size : l.size); }   // Activation::sizes

struct Episode {  // ReplayMemory/Episode.h:40-108
  int64_t tag = -1, ID = -1, seq = 0; int N = 0; bool term = false;
  std::vector<float> S; std::vector<double> A, MU, R;
  std::vector<nnReal> V, ADV, RET;         // stateValue, actionAdvantage, returnEstimator
  std::vector<Fval> DQ, IMPW, DKL;         // deltaValue, offPolicImpW, KullbLeibDiv
  Fval totR = 0, avgKL = 0, fracFar = 0, avgSqErr = 0, maxAbsErr = 0;
  Fval sumQ2 = 0, sumQ = 0, maxQ = -1e9, minQ = 1e9;
  int ndata() const { return N - 1; }
  bool isTruncated(int t) const { return t + 1 == N && !term; }
  bool isTerminal(int t) const { return t + 1 == N && term; }
};

}  // namespace

struct ol_learner {
  hl_config cfg;
  std::string err;
  int dS = 0, dA = 0, nOut = 0, B = 0, Bglobal = 0;
  int64_t maxObsLocal = 0, maxObsGlobal = 0, minObsLocal = 0;
  std::vector<Layer> layers;
  int nEncLayers = 0;           // hidden layers that came from encoderLayerSizes (the first ones of the merged list)
  int64_t nParams = 0;
  std::vector<nnReal> W, M1, M2, G;
  MT19937 gen;
  // replay
  std::vector<std::unique_ptr<Episode>> episodes;
  std::vector<nnReal> stMean, stStd, stScale; nnReal rewMean = 0, rewStd = 1, rewScale = 1;
  Real beta = 1e-4, alpha = 0.5, CmaxRet = 5, CinvRet = 0.25;
  int64_t nGradSteps = 0, nTransitions = 0, nSeenSteps = 0, nSeenEps = 0;
  // ReplayCounters::nSeenEpisodes / nSeenTransitions: the (summed) counters as of the last updateCounters (MemoryProcessing.cpp:60-61)
  // -- what nSeenEps() / nSeenSteps() return and the stats line prints; the live ones above are the reference's *_loc
  int64_t seenEpsUpd = 0, seenStepsUpd = 0, seenEpsGlobal = 0, seenStepsGlobal = 0;
  int64_t nGatheredB4Startup = INT64_MAX;
  int64_t nFarGlobal = 0, nStoredGlobal = 0;  // result of counters reduction
  bool countersReduced = false, momentsPending = false, initPending = false;
  std::vector<long double> moments;
  hl_stats stats{};
  // adam (Network/Optimizer.h:38-47,96)
  Real beta_t_1 = 0.9, beta_t_2 = 0.999; int64_t nStep = 0;
  int nAdv = 0;                 // advantage outputs between V and the policy mean (0 = VRACER)
  int nOpt = 0;                 // discrete head: number of options (0 = continuous)
  int polDim = 0;               // entries of a stored behaviour policy: 2 dA (mean | stdev) or nOpt probabilities
  bool initialized = false, inStep = false, tap = false;
  // activations workspace: per layer X, Y, E for current and next step
  std::vector<std::vector<nnReal>> X, Y, E, Xn, Yn;
  // last batch
  std::vector<int64_t> bFlat, bEp, bT, bTag;
  std::vector<float> tState; std::vector<double> tO, tG, tRho, tDkl, tDq; std::vector<uint8_t> tFar;
  std::vector<nnReal> tGradSum;
  // Utils/StatsTracker.cpp: sums of the output gradients over the minibatch, mean / RMS of the last one
  std::vector<long double> gsSum, gsSq; std::vector<double> gsMean, gsRms;
  int64_t gsCalls = 0;           // StatsTracker::nStep
  std::string logBase;           // "<learner_name>": <logBase>_net_outGrad_stats.raw
  std::string episodeLog;        // cumulative_rewards.dat (MemoryBuffer.cpp:481-507)
};

namespace {

int fail(ol_learner* h, int code, const std::string& msg) { if (h) h->err = msg; return code; }

// Builder::addInput/addLayer/addParamLayer (Network/Builder.cpp:26-117) driven by
// Approximator::buildFromSettings (Network/Approximator.cpp:179-229) and
// RACER::setupNet (Learners/RACER_common.cpp:71-115) with RACER_simpleSigma.
void buildNet(ol_learner* h) {
  const hl_config& c = h->cfg;
  std::vector<Layer>& L = h->layers;
  L.clear();
  // input: the observed state followed by the nAppendedObs previous ones (Approximator.cpp:208-209, 242-243)
  // with convolutions the first input layer is the first convolution's image; state variables beyond it become a second
  // input layer behind the conv stack, glued to it by a JoinLayer (Approximator.cpp:245-259, Builder.cpp:26-46)
  const int inAll = c.dimS * (1 + c.nAppendedObs);
  const int inImg = c.n_conv > 0 ? c.conv[0].inpFeatures * c.conv[0].inpY * c.conv[0].inpX : inAll;
  { Layer in; in.type = L_INPUT; in.size = inImg; L.push_back(in); }
  // Approximator::buildPreprocessing (Approximator.cpp:231-271) -> Builder::addConv2d (Builder.cpp:172-215): SoftSign
  // convolutions, no skip connections between them
  for (int j = 0; j < c.n_conv; ++j) {
    const hl_conv2d& d = c.conv[j];
    Layer cl; cl.type = L_CONV; cl.cv = d; cl.func = HL_FUNC_SOFTSIGN;
    cl.size = d.outFeatures * d.outY * d.outX; cl.nIn = d.inpFeatures * d.inpY * d.inpX;
    L.push_back(cl);
  }
  if (inAll > inImg) {
    Layer in; in.type = L_INPUT; in.size = inAll - inImg; L.push_back(in);
    Layer jn; jn.type = L_JOIN; jn.size = in.size + L[L.size() - 2].size; L.push_back(jn);      // JoinLayer(ID + 1, twoLayersSize, 2)
  }
  size_t nHid = 0;
  for (int j = 0; j < c.n_hidden; ++j) {
    if (c.hidden[j] <= 0) continue;
    const int ID = (int)L.size();
    Layer d; d.type = c.nn_type == HL_NN_LSTM ? L_LSTM : (c.nn_type == HL_NN_MGU ? L_MGU : L_DENSE); d.size = c.hidden[j]; d.nIn = L[ID - 1].size;
    d.rec = c.nn_type == HL_NN_RNN;
    if (c.encoder_rnn && (int)nHid < h->nEncLayers) { d.type = L_DENSE; d.rec = true; }      // "RNN" encoder layers of a partially observable MDP (Approximator.cpp:264-270)
    ++nHid;
    d.nOutSimd = (int)roundUp8(d.size); d.func = c.nnFunc;
    L.push_back(d);
    // skip connection except after the first layer (Builder.cpp:89-95)
    if (ID != 1) { Layer r; r.type = L_PARAMRES; r.size = d.size; L.push_back(r); }
  }
  // VRACER: [V, mean x dA]; RACER with the Gaussian advantage: [V, coef, L+ x dA, L- x dA, mean x dA]
  // (RACER_common.cpp:172-186, Gaus_advantage.h:20-22); sigma is a ParamLayer (RACER_simpleSigma)
  // RACER discrete: [V, A x nOpt, logits x nOpt], no sigma layer (RACER_common.cpp:119-134)
  const bool discrete = c.adv_kind == HL_ADV_DISCRETE;
  const int nAdv = c.adv_kind == HL_ADV_GAUSSIAN ? 1 + 2 * c.dimA : (discrete ? c.n_options : 0);
  const int nDense = 1 + nAdv + (discrete ? c.n_options : c.dimA);
  h->nAdv = nAdv; h->nOpt = discrete ? c.n_options : 0; h->polDim = discrete ? c.n_options : 2 * c.dimA;
  { const int ID = (int)L.size();
    Layer o; o.type = L_DENSE; o.size = nDense; o.nIn = L[ID - 1].size;
    o.nOutSimd = (int)roundUp8(o.size); o.func = c.nnOutputFunc; o.bOutput = true;   // settings nnOutputFunc (Approximator.cpp:193,228)
    // continuous actions: Builder::setLastLayersBias(biases) with {0 (V) | Advantage_t::setInitial | Policy_t::setInitial_noStdev
    // = zeros} (RACER_common.cpp:94-105); the layer stores the pre-images under its function.  Discrete: biases stay zero.
    if (!discrete) o.biasInit.assign(nDense, 0);
    if (c.adv_kind == HL_ADV_GAUSSIAN) {   // Gaussian_advantage::setInitial (Gaus_advantage.h:31-34)
      o.biasInit[1] = -1;
      for (int e = 2; e < 1 + nAdv; ++e) o.biasInit[e] = 1;
    }
    L.push_back(o); }
  if (!discrete) { Layer p; p.type = L_PARAM; p.size = c.dimA; p.func = HL_FUNC_LINEAR; p.bOutput = true;
    // Continuous_policy::initial_Stdev -> SoftPlus::_inv(explNoise) (Continuous_policy.h:603-617,192-194)
    Real S = c.explNoise; if (S < FLT_EPSILON) S = FLT_EPSILON;
    p.biasInit.assign(c.dimA, spInv(S)); L.push_back(p); }
  // layer 1 never back-propagates to the input vector (Approximator.cpp:146-170)
  if (L.size() > 1) L[1].skipInpGrad = true;
  // Parameters::_computeNParams (Layers/Parameters.h:159-176)
  int64_t tot = 0;
  for (auto& l : L) {
    switch (l.type) {
      case L_INPUT: case L_JOIN: l.nW = 0; l.nB = 0; break;
      case L_DENSE: l.nW = (int64_t)l.nOutSimd * (l.nIn + (l.rec ? l.size : 0)); l.nB = l.size; break;   // Layer_Base.h:24-28
      case L_PARAMRES: l.nW = l.size; l.nB = l.size; break;                    // Layers.h:334-338
      case L_PARAM: l.nW = 0; l.nB = l.size; break;                            // Layers.h:494-497
      case L_LSTM: l.nW = (int64_t)4 * l.size * (l.nIn + l.size); l.nB = 4 * l.size; break;   // Layer_LSTM.h:24-29
      case L_MGU: l.nW = (int64_t)2 * l.size * (l.nIn + l.size); l.nB = 2 * l.size; break;    // Layer_GRU.h:29-34
      case L_CONV: l.nW = (int64_t)l.cv.outFeatures * l.cv.inpFeatures * l.cv.filtery * l.cv.filterx; l.nB = l.size; break;   // Layer_Conv2D.h:37-40
    }
    l.indW = tot; tot += roundUp8(l.nW);
    l.indB = tot; tot += roundUp8(l.nB);
  }
  h->nParams = tot;
  h->nOut = nDense + (discrete ? 0 : c.dimA);
  h->W.assign(tot, 0); h->M1.assign(tot, 0); h->M2.assign(tot, 0); h->G.assign(tot, 0);
  const size_t nl = L.size();
  h->X.resize(nl); h->Y.resize(nl); h->E.resize(nl); h->Xn.resize(nl); h->Yn.resize(nl);
  for (size_t i = 0; i < nl; ++i) {
    const size_t n = (size_t)roundUp8(actSize(L[i]));
    h->X[i].assign(n, 0); h->Y[i].assign(n, 0); h->E[i].assign(n, 0);
    h->Xn[i].assign(n, 0); h->Yn[i].assign(n, 0);
  }
}

// Layer::initialize in build order (Builder.cpp:131-137; Layer_Base.h:115-141;
// Layers.h:395-400, 548-553)
void initWeights(ol_learner* h) {
  const hl_config& c = h->cfg;
  for (const Layer& l : h->layers) {
    nnReal* W = h->W.data() + l.indW; nnReal* Bv = h->W.data() + l.indB;
    if (l.type == L_DENSE) {
      const Real initializationFac = l.bOutput ? c.outWeightsPrefac : 1;
      const nnReal fac = (initializationFac > 0) ? initializationFac : 1;
      const nnReal init = fac * fInitFactor(l.func, l.nIn, l.size);
      for (int o = 0; o < l.size; ++o) Bv[o] = l.biasInit.size() == (size_t)l.size ? fInverse(l.func, (nnReal)l.biasInit[o]) : 0;  // pre-image of the init values (Layer_Base.h:122-125)
      for (int i = 0; i < l.nIn + (l.rec ? l.size : 0); ++i)      // input weights, then the recurrent ones (:127-140)
        for (int o = 0; o < l.size; ++o) W[o + (int64_t)l.nOutSimd * i] = uniformFloat(h->gen, -init, init);
    } else if (l.type == L_LSTM) {   // Layer_LSTM.h:167-185: forget gate starts open, input / output gates closed (LSTM_PRIME_FAC = 1)
      const nnReal init = fInitFactor(l.func, l.nIn, l.size);
      const int nC = l.size;
      for (int o = 0; o < nC; ++o) { Bv[o] = 0; Bv[nC + o] = -1; Bv[2 * nC + o] = 1; Bv[3 * nC + o] = -1; }
      for (int64_t w = 0; w < (int64_t)4 * nC * (l.nIn + nC); ++w) W[w] = uniformFloat(h->gen, -init, init);
    } else if (l.type == L_MGU) {    // Layer_GRU.h:232-246: forget gate starts open
      const nnReal init = fInitFactor(l.func, l.nIn, l.size);
      const int nC = l.size;
      for (int o = 0; o < nC; ++o) { Bv[o] = 1; Bv[nC + o] = 0; }
      for (int64_t w = 0; w < (int64_t)2 * nC * (l.nIn + nC); ++w) W[w] = uniformFloat(h->gen, -init, init);
    } else if (l.type == L_PARAMRES) {
      for (int o = 0; o < l.size; ++o) { Bv[o] = 0; W[o] = 1; }
    } else if (l.type == L_PARAM) {
      for (int o = 0; o < l.size; ++o) Bv[o] = (nnReal)l.biasInit[o];
    } else if (l.type == L_CONV) {   // Layer_Conv2D.h:198-213: fan-in InC KnX KnY, fan-out KnC; biases zero, weights in memory order
      const nnReal init = fInitFactor(l.func, l.cv.inpFeatures * l.cv.filterx * l.cv.filtery, l.cv.outFeatures);
      for (int o = 0; o < l.size; ++o) Bv[o] = 0;
      for (int64_t w = 0; w < l.nW; ++w) W[w] = uniformFloat(h->gen, -init, init);
    }
  }
}

// Conv2DLayer::forward / backward, the direct loops of the build without BLAS (Layer_Conv2D.h:83-139): image [C][Y][X],
// filter K[KnC][InC][KnY][KnX], the reference's loop nest (so every output element sums its terms in the same order)
void convForward(const Layer& l, const nnReal* K, const nnReal* Bv, const nnReal* INP, nnReal* OUT, nnReal* Yout) {
  const hl_conv2d& d = l.cv;
  const int InC = d.inpFeatures, InY = d.inpY, InX = d.inpX, KnC = d.outFeatures, KnY = d.filtery, KnX = d.filterx;
  const int OpY = d.outY, OpX = d.outX, Sx = d.stridex, Sy = d.stridey, Px = d.paddinx, Py = d.paddiny;
  std::memcpy(OUT, Bv, (size_t)l.size * sizeof(nnReal));
  for (int fc = 0; fc < KnC; ++fc) for (int ic = 0; ic < InC; ++ic)
  for (int oy = 0; oy < OpY; ++oy) for (int fy = 0; fy < KnY; ++fy)
  for (int ox = 0; ox < OpX; ++ox) for (int fx = 0; fx < KnX; ++fx) {
    const int ix = ox * Sx - Px + fx, iy = oy * Sy - Py + fy;
    if (ix < 0 || ix >= InX || iy < 0 || iy >= InY) continue;
    OUT[(fc * OpY + oy) * OpX + ox] += K[((fc * InC + ic) * KnY + fy) * KnX + fx] * INP[(ic * InY + iy) * InX + ix];
  }
  for (int o = 0; o < l.size; ++o) Yout[o] = fEval(l.func, OUT[o]);
}
void convBackward(const Layer& l, const nnReal* K, const nnReal* INP, const nnReal* Xl, const nnReal* Yl, nnReal* dOUT,
                  nnReal* dINP /*nullptr: first layer*/, nnReal* gK, nnReal* gB) {
  const hl_conv2d& d = l.cv;
  const int InC = d.inpFeatures, InY = d.inpY, InX = d.inpX, KnC = d.outFeatures, KnY = d.filtery, KnX = d.filterx;
  const int OpY = d.outY, OpX = d.outX, Sx = d.stridex, Sy = d.stridey, Px = d.paddinx, Py = d.paddiny;
  for (int o = 0; o < l.size; ++o) { dOUT[o] *= fDiff(l.func, Xl[o], Yl[o]); gB[o] += dOUT[o]; }   // backward_bias (:68-78)
  for (int fc = 0; fc < KnC; ++fc) for (int ic = 0; ic < InC; ++ic)
  for (int oy = 0; oy < OpY; ++oy) for (int fy = 0; fy < KnY; ++fy)
  for (int ox = 0; ox < OpX; ++ox) for (int fx = 0; fx < KnX; ++fx) {
    const int ix = ox * Sx - Px + fx, iy = oy * Sy - Py + fy;
    if (ix < 0 || ix >= InX || iy < 0 || iy >= InY) continue;
    const nnReal dO = dOUT[(fc * OpY + oy) * OpX + ox];
    gK[((fc * InC + ic) * KnY + fy) * KnX + fx] += dO * INP[(ic * InY + iy) * InX + ix];
    if (dINP) dINP[(ic * InY + iy) * InX + ix] += dO * K[((fc * InC + ic) * KnY + fy) * KnX + fx];
  }
}

// Network::forward (Network/Network.h:102-113) over Layer::forward of each type
inline nnReal sigmEval(nnReal in) {   // Sigm::_eval (Functions.h:158-165), safeExp cut at 8 for fp32 (Definitions.h:43)
  const auto safeExp = [](nnReal v) { return std::exp(std::min((nnReal)8, std::max(-(nnReal)8, v))); };
  if (in > 0) return 1 / (1 + safeExp(-in));
  const nnReal ex = safeExp(in); return ex / (1 + ex);
}
void forwardNet(const ol_learner* h, const nnReal* input, std::vector<std::vector<nnReal>>& X,
                std::vector<std::vector<nnReal>>& Y, const std::vector<std::vector<nnReal>>* prevY = nullptr) {
  const auto& L = h->layers;
  { int k = 0;      // Activation::setInput (Layers/Activation.h:58-70): the input vector, cut over the input layers in order
    for (size_t ID = 0; ID < L.size(); ++ID) if (L[ID].type == L_INPUT) { std::copy(input + k, input + k + L[ID].size, Y[ID].begin()); k += L[ID].size; } }
  for (size_t ID = 1; ID < L.size(); ++ID) {
    const Layer& l = L[ID];
    if (l.type == L_INPUT) continue;
    if (l.type == L_JOIN) {      // JoinLayer::forward (Layers.h:289-299): the layer before first, then the one before that
      const int n1 = L[ID - 1].size, n2 = L[ID - 2].size;
      std::copy(Y[ID - 1].begin(), Y[ID - 1].begin() + n1, Y[ID].begin());
      std::copy(Y[ID - 2].begin(), Y[ID - 2].begin() + n2, Y[ID].begin() + n1);
      continue;
    }
    const nnReal* W = h->W.data() + l.indW; const nnReal* Bv = h->W.data() + l.indB;
    if (l.type == L_DENSE) {  // Layer_Base.h:64-95
      nnReal* suminp = X[ID].data();
      std::memcpy(suminp, Bv, l.size * sizeof(nnReal));
      const nnReal* inputs = Y[ID - 1].data();
      for (int i = 0; i < l.nIn; ++i) {
        const nnReal* Wi = W + (int64_t)l.nOutSimd * i;
        for (int o = 0; o < l.size; ++o) suminp[o] += inputs[i] * Wi[o];
      }
      if (l.rec && prevY) {      // Layer_Base.h:83-93: + W_rec y_{t-1}
        const nnReal* rin = (*prevY)[ID].data(); const nnReal* Wr = W + (int64_t)l.nOutSimd * l.nIn;
        for (int i = 0; i < l.size; ++i) { const nnReal* Wi = Wr + (int64_t)l.nOutSimd * i; for (int o = 0; o < l.size; ++o) suminp[o] += rin[i] * Wi[o]; }
      }
      for (int o = 0; o < l.size; ++o) Y[ID][o] = fEval(l.func, suminp[o]);
    } else if (l.type == L_PARAMRES) {  // Layers.h:347-361
      nnReal* ret = Y[ID].data();
      std::memcpy(ret, Y[ID - 1].data(), l.size * sizeof(nnReal));
      const nnReal* inp = Y[ID - 2].data();
      const int sizeInp = std::min(actSize(L[ID - 2]), l.size);
      for (int j = 0; j < sizeInp; ++j) ret[j] += inp[j] * W[j] + Bv[j];
    } else if (l.type == L_LSTM) {   // Layer_LSTM.h:78-125
      const int nC = l.size;
      nnReal* suminp = X[ID].data();
      std::memcpy(suminp, Bv, 4 * nC * sizeof(nnReal));
      { const nnReal* inputs = Y[ID - 1].data();
        for (int i = 0; i < l.nIn; ++i) { const nnReal* Wi = W + (int64_t)4 * nC * i; for (int o = 0; o < 4 * nC; ++o) suminp[o] += inputs[i] * Wi[o]; } }
      if (prevY) {
        const nnReal* inputs = (*prevY)[ID].data(); const nnReal* Wr = W + (int64_t)4 * nC * l.nIn;
        for (int i = 0; i < nC; ++i) { const nnReal* Wi = Wr + (int64_t)4 * nC * i; for (int o = 0; o < 4 * nC; ++o) suminp[o] += inputs[i] * Wi[o]; }
      }
      for (int o = nC; o < 4 * nC; ++o) suminp[o] = sigmEval(suminp[o]);      // gates overwrite their inputs
      const nnReal* prevSt = prevY ? (*prevY)[ID].data() + nC : nullptr;
      nnReal* output = Y[ID].data(); nnReal* currSt = output + nC; nnReal* cellOp = output + 2 * nC;
      const nnReal* inputG = suminp + nC; const nnReal* forgtG = suminp + 2 * nC; const nnReal* outptG = suminp + 3 * nC;
      for (int o = 0; o < nC; ++o) {
        const nnReal oldStatePass = prevSt ? prevSt[o] * forgtG[o] : 0;
        currSt[o] = suminp[o] * inputG[o] + oldStatePass;
        cellOp[o] = fEval(HL_FUNC_TANH, currSt[o]);
        output[o] = outptG[o] * cellOp[o];
      }
    } else if (l.type == L_MGU) {    // Layer_GRU.h:66-118: forget = sigm(Wff in + Wfr prevOut + bf), state = tanh(Wsf in + Wsr (forget*prevOut) + bs)
      const int nC = l.size;
      nnReal* forget = X[ID].data(); nnReal* state = forget + nC; nnReal* output = Y[ID].data();
      std::memcpy(forget, Bv, 2 * nC * sizeof(nnReal));
      { const nnReal* inputs = Y[ID - 1].data();
        for (int i = 0; i < l.nIn; ++i) { const nnReal* Wi = W + (int64_t)2 * nC * i; for (int o = 0; o < 2 * nC; ++o) forget[o] += inputs[i] * Wi[o]; } }
      if (prevY) {
        const nnReal* inputs = (*prevY)[ID].data(); const nnReal* Wr = W + (int64_t)2 * nC * l.nIn;
        for (int i = 0; i < nC; ++i) { const nnReal* Wfr = Wr + (int64_t)2 * nC * i; for (int o = 0; o < nC; ++o) forget[o] += Wfr[o] * inputs[i]; }
        for (int o = 0; o < nC; ++o) forget[o] = sigmEval(forget[o]);
        for (int i = 0; i < nC; ++i) { const nnReal* Wsr = Wr + (int64_t)2 * nC * i + nC; for (int o = 0; o < nC; ++o) state[o] += Wsr[o] * inputs[i] * forget[i]; }
        for (int o = 0; o < nC; ++o) state[o] = fEval(HL_FUNC_TANH, state[o]);
        for (int o = 0; o < nC; ++o) output[o] = forget[o] * state[o] + (1 - forget[o]) * inputs[o];
      } else {
        for (int o = 0; o < nC; ++o) forget[o] = sigmEval(forget[o]);
        for (int o = 0; o < nC; ++o) state[o] = fEval(HL_FUNC_TANH, state[o]);
        for (int o = 0; o < nC; ++o) output[o] = forget[o] * state[o];
      }
    } else if (l.type == L_PARAM) {  // Layers.h:510-520
      for (int n = 0; n < l.size; ++n) { X[ID][n] = Bv[n]; Y[ID][n] = fEval(l.func, Bv[n]); }
    } else if (l.type == L_CONV) {
      convForward(l, W, Bv, Y[ID - 1].data(), X[ID].data(), Y[ID].data());
    }
  }
}

// Activation::getOutput (Layers/Activation.h:134-148): output layers in order
void getOutput(const ol_learner* h, const std::vector<std::vector<nnReal>>& Y, Real* O) {
  int k = 0;
  for (size_t i = 0; i < h->layers.size(); ++i)
    if (h->layers[i].bOutput) for (int j = 0; j < h->layers[i].size; ++j) O[k++] = Y[i][j];
}

// GEMVomp (Layers/Layers.h:34-60): errors[o] += sum_i W[o*S+i]*deltas[i], blocked by 16
void gemvOmp(int NX, int NY, int S, const nnReal* Wm, const nnReal* Xv, nnReal* Yv) {
  constexpr int cacheLineLen = 64 / sizeof(nnReal);
  for (int I = 0; I < NX; I += cacheLineLen)
    for (int o = 0; o < NY; ++o) {
      const nnReal* Wr = Wm + (int64_t)S * o;
      nnReal acc = 0;
      const int Ninner = std::min(NX, I + cacheLineLen);
      for (int i = I; i < Ninner; ++i) acc += Wr[i] * Xv[i];
      Yv[o] += acc;
    }
}

// Network::backProp single step (Network/Network.h:155-166 -> :220-231) with the
// per-type backward (Layers.h:123-188, 363-393, 522-546; Layer_Base.h:97-113).
// E[] must hold the output deltas and zeros elsewhere; accumulates into G.
void backwardNet(ol_learner* h) {
  const auto& L = h->layers;
  auto& X = h->X; auto& Y = h->Y; auto& E = h->E;
  for (int ID = (int)L.size() - 1; ID >= 1; --ID) {
    const Layer& l = L[ID];
    const nnReal* W = h->W.data() + l.indW;
    nnReal* gW = h->G.data() + l.indW; nnReal* gB = h->G.data() + l.indB;
    if (l.type == L_PARAM) {
      nnReal* deltas = E[ID].data();
      for (int o = 0; o < l.size; ++o) { deltas[o] *= fDiff(l.func, X[ID][o], Y[ID][o]); gB[o] += deltas[o]; }
    } else if (l.type == L_DENSE) {
      nnReal* deltas = E[ID].data();
      for (int o = 0; o < l.size; ++o) deltas[o] *= fDiff(l.func, X[ID][o], Y[ID][o]);
      if (!l.skipInpGrad) gemvOmp(l.size, l.nIn, l.nOutSimd, W, deltas, E[ID - 1].data());
      for (int o = 0; o < l.size; ++o) gB[o] += deltas[o];
      const nnReal* inputs = Y[ID - 1].data();
      for (int i = 0; i < l.nIn; ++i) {
        nnReal* Gi = gW + (int64_t)l.nOutSimd * i;
        for (int o = 0; o < l.size; ++o) Gi[o] += inputs[i] * deltas[o];
      }
    } else if (l.type == L_PARAMRES) {
      const nnReal* delta = E[ID].data();
      std::memcpy(E[ID - 1].data(), delta, l.size * sizeof(nnReal));
      nnReal* gradInp = E[ID - 2].data();
      const nnReal* inp = Y[ID - 2].data();
      const int sizeInp = std::min(L[ID - 2].size, l.size);
      for (int j = 0; j < sizeInp; ++j) {
        gradInp[j] += delta[j] * W[j];
        gW[j] += delta[j] * inp[j];
        gB[j] += delta[j];
      }
    } else if (l.type == L_CONV) {
      convBackward(l, W, Y[ID - 1].data(), X[ID].data(), Y[ID].data(), E[ID].data(), ID > 1 ? E[ID - 1].data() : nullptr, gW, gB);
    } else if (l.type == L_JOIN) {      // JoinLayer::backward (Layers.h:301-313): errors handed back, overwriting
      const int n1 = L[ID - 1].size, n2 = L[ID - 2].size;
      std::copy(E[ID].begin(), E[ID].begin() + n1, E[ID - 1].begin());
      std::copy(E[ID].begin() + n1, E[ID].begin() + n1 + n2, E[ID - 2].begin());
    }
  }
}

// Network::backProp over a time series (Network/Network.h:155-193): layers from the top down, for each layer the
// steps from the last one with an error (T) back to 0; recurrent layers pass errors to the previous step's E
void backwardSeries(ol_learner* h, std::vector<Act>& series, int T) {
  const auto& L = h->layers;
  for (int ID = (int)L.size() - 1; ID >= 1; --ID) {
    const Layer& l = L[ID];
    const nnReal* W = h->W.data() + l.indW;
    nnReal* gW = h->G.data() + l.indW; nnReal* gB = h->G.data() + l.indB;
    for (int k = T; k >= 0; --k) {
      Act& cur = series[k]; Act* prev = k > 0 ? &series[k - 1] : nullptr; Act* next = k < T ? &series[k + 1] : nullptr;
      if (l.type == L_PARAM) {
        nnReal* deltas = cur.E[ID].data();
        for (int o = 0; o < l.size; ++o) { deltas[o] *= fDiff(l.func, cur.X[ID][o], cur.Y[ID][o]); gB[o] += deltas[o]; }
      } else if (l.type == L_DENSE) {
        nnReal* deltas = cur.E[ID].data();
        for (int o = 0; o < l.size; ++o) deltas[o] *= fDiff(l.func, cur.X[ID][o], cur.Y[ID][o]);
        if (!l.skipInpGrad) gemvOmp(l.size, l.nIn, l.nOutSimd, W, deltas, cur.E[ID - 1].data());
        if (l.rec && prev) gemvOmp(l.size, l.size, l.nOutSimd, W + (int64_t)l.nOutSimd * l.nIn, deltas, prev->E[ID].data());   // Layers.h:148-159
        for (int o = 0; o < l.size; ++o) gB[o] += deltas[o];
        const nnReal* inputs = cur.Y[ID - 1].data();
        for (int i = 0; i < l.nIn; ++i) { nnReal* Gi = gW + (int64_t)l.nOutSimd * i; for (int o = 0; o < l.size; ++o) Gi[o] += inputs[i] * deltas[o]; }
        if (l.rec && prev) { const nnReal* rin = prev->Y[ID].data(); nnReal* gR = gW + (int64_t)l.nOutSimd * l.nIn;                  // :178-187
          for (int i = 0; i < l.size; ++i) { nnReal* Gi = gR + (int64_t)l.nOutSimd * i; for (int o = 0; o < l.size; ++o) Gi[o] += rin[i] * deltas[o]; } }
      } else if (l.type == L_PARAMRES) {
        const nnReal* delta = cur.E[ID].data();
        std::memcpy(cur.E[ID - 1].data(), delta, l.size * sizeof(nnReal));
        nnReal* gradInp = cur.E[ID - 2].data(); const nnReal* inp = cur.Y[ID - 2].data();
        const int sizeInp = std::min(actSize(L[ID - 2]), l.size);
        for (int j = 0; j < sizeInp; ++j) { gradInp[j] += delta[j] * W[j]; gW[j] += delta[j] * inp[j]; gB[j] += delta[j]; }
      } else if (l.type == L_CONV) {      // (a conv stack in front of recurrent layers: every step of the window passes through it)
        convBackward(l, W, cur.Y[ID - 1].data(), cur.X[ID].data(), cur.Y[ID].data(), cur.E[ID].data(), ID > 1 ? cur.E[ID - 1].data() : nullptr, gW, gB);
      } else if (l.type == L_JOIN) {
        const int n1 = L[ID - 1].size, n2 = L[ID - 2].size;
        std::copy(cur.E[ID].begin(), cur.E[ID].begin() + n1, cur.E[ID - 1].begin());
        std::copy(cur.E[ID].begin() + n1, cur.E[ID].begin() + n1 + n2, cur.E[ID - 2].begin());
      } else if (l.type == L_MGU) {    // Layer_GRU.h:120-229
        const int nC = l.size;
        const nnReal* forget = cur.X[ID].data(); const nnReal* state = forget + nC;
        const nnReal* dLdO = cur.E[ID].data(); nnReal* dLdF = cur.E[ID].data() + nC; nnReal* dLdS = cur.Y[ID].data() + nC;
        std::vector<nnReal> zeros(nC, 0), dLdFprevOut(nC, 0);
        const nnReal* prevOut = prev ? prev->Y[ID].data() : zeros.data();
        nnReal* dLdprevOut = prev ? prev->E[ID].data() : nullptr;
        for (int o = 0; o < nC; ++o) dLdS[o] = dLdO[o] * forget[o] * (1 - state[o] * state[o]);
        const nnReal* Wr = W + (int64_t)2 * nC * l.nIn;
        if (prev) gemvOmp(nC, nC, 2 * nC, Wr + nC, dLdS, dLdFprevOut.data());
        for (int o = 0; o < nC; ++o) dLdF[o] = ((state[o] - prevOut[o]) * dLdO[o] + dLdFprevOut[o] * prevOut[o]) * forget[o] * (1 - forget[o]);
        if (prev) {
          for (int o = 0; o < nC; ++o) dLdprevOut[o] += (1 - forget[o]) * dLdO[o] + forget[o] * dLdFprevOut[o];
          gemvOmp(nC, nC, 2 * nC, Wr, dLdF, dLdprevOut);
        }
        if (!l.skipInpGrad) {
          gemvOmp(nC, l.nIn, 2 * nC, W, dLdF, cur.E[ID - 1].data());
          gemvOmp(nC, l.nIn, 2 * nC, W + nC, dLdS, cur.E[ID - 1].data());
        }
        for (int o = 0; o < nC; ++o) { gB[o] += dLdF[o]; gB[o + nC] += dLdS[o]; }
        { const nnReal* inputs = cur.Y[ID - 1].data();
          for (int i = 0; i < l.nIn; ++i) { nnReal* Gi = gW + (int64_t)2 * nC * i; for (int o = 0; o < nC; ++o) { Gi[o] += inputs[i] * dLdF[o]; Gi[o + nC] += inputs[i] * dLdS[o]; } } }
        if (prev) for (int i = 0; i < nC; ++i) {
          nnReal* Gi = gW + (int64_t)2 * nC * (l.nIn + i);
          for (int o = 0; o < nC; ++o) { Gi[o] += prevOut[i] * dLdF[o]; Gi[o + nC] += prevOut[i] * dLdS[o] * forget[i]; }
        }
      } else if (l.type == L_LSTM) {   // Layer_LSTM.h:127-165, then Layer::backward (Layers.h:123-188) with NO = 4 nC, NR = nC
        const int nC = l.size;
        nnReal* deltas = cur.E[ID].data();
        const nnReal* cellOutput = cur.Y[ID].data() + 2 * nC; nnReal* stateDelta = cur.Y[ID].data() + 3 * nC;
        const nnReal* cellInpt = cur.X[ID].data(); const nnReal* IGate = cellInpt + nC; const nnReal* FGate = cellInpt + 2 * nC;
        const nnReal* OGate = cellInpt + 3 * nC;
        const nnReal* prvState = prev ? prev->Y[ID].data() + nC : nullptr;
        const nnReal* nxtStErr = next ? next->Y[ID].data() + 3 * nC : nullptr;
        const nnReal* nxtFGate = next ? next->X[ID].data() + 2 * nC : nullptr;
        for (int o = 0; o < nC; ++o) {
          const nnReal D = deltas[o];
          const nnReal diff = (1 - cellOutput[o] * cellOutput[o]) * deltas[o];
          stateDelta[o] = diff * OGate[o] + (next ? nxtStErr[o] * nxtFGate[o] : 0);
          deltas[o] = IGate[o] * stateDelta[o];
          deltas[o + nC] = IGate[o] * (1 - IGate[o]) * cellInpt[o] * stateDelta[o];
          deltas[o + 2 * nC] = prev ? FGate[o] * (1 - FGate[o]) * prvState[o] * stateDelta[o] : 0;
          deltas[o + 3 * nC] = OGate[o] * (1 - OGate[o]) * D * cellOutput[o];
        }
        const int NO = 4 * nC;
        if (!l.skipInpGrad) gemvOmp(NO, l.nIn, NO, W, deltas, cur.E[ID - 1].data());
        if (prev) gemvOmp(NO, nC, NO, W + (int64_t)NO * l.nIn, deltas, prev->E[ID].data());
        for (int o = 0; o < NO; ++o) gB[o] += deltas[o];
        { const nnReal* inputs = cur.Y[ID - 1].data();
          for (int i = 0; i < l.nIn; ++i) { nnReal* Gi = gW + (int64_t)NO * i; for (int o = 0; o < NO; ++o) Gi[o] += inputs[i] * deltas[o]; } }
        if (prev) { const nnReal* inputs = prev->Y[ID].data(); nnReal* gR = gW + (int64_t)NO * l.nIn;
          for (int i = 0; i < nC; ++i) { nnReal* Gi = gR + (int64_t)NO * i; for (int o = 0; o < NO; ++o) Gi[o] += inputs[i] * deltas[o]; } }
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// RACER::Train head for VRACER (Learners/RACER_train.cpp:31-60) with
// Continuous_policy (Math/Continuous_policy.h:569-738): NormalPolicy (:68-210)
// for unbounded, SquashedNormalPolicy (:212-378) for bounded components.
// ---------------------------------------------------------------------------
extern "C" void ol_head_vracer(int dA, const uint8_t* bounded, const double* O, const double* act,
                               const double* mu, double Qret, double beta, double Cmax, double Cinv,
                               double* grad, double* rho, double* dkl, double* deltaQ, int* isFar,
                               double* Vval) {
  double Q;
  ol_head_racer(dA, 0, bounded, O, act, mu, Qret, beta, Cmax, Cinv, grad, rho, dkl, deltaQ, isFar, Vval, &Q);
}
// nAdv = 0: Zero_advantage (VRACER); nAdv = 1 + 2 dA: Gaussian_advantage (Math/Gaus_advantage.h:17-127).
// Output layout [V | advantage nAdv | mean dA | sigma parameter dA].
extern "C" void ol_head_racer(int dA, int nAdv, const uint8_t* bounded, const double* O, const double* act,
                              const double* mu, double Qret, double beta, double Cmax, double Cinv,
                              double* grad, double* rho, double* dkl, double* deltaQ, int* isFar,
                              double* Vval, double* Qval) {
  const int pM = 1 + nAdv, pS = 1 + nAdv + dA;      // policy mean / sigma-parameter offsets
  for (int o = 0; o < pS + dA; ++o) grad[o] = 0;
  constexpr Real MAXM = 8.31776613503286;
  constexpr Real LOG2PI_2 = 9.1893853320467266954096885456237942e-01;
  constexpr Real FMIN = FLT_MIN;
  Real logW = 0, kl = 0;
  std::vector<Real> mean(dA), stdev(dA), invStd(dA), dPos(dA);
  for (int i = 0; i < dA; ++i) {
    mean[i] = O[pM + i];
    const Real p = O[pS + i];
    stdev[i] = spEval(p); invStd[i] = 1 / stdev[i]; dPos[i] = spDiff(p);
    const Real bMean = mu[i], bStd = mu[dA + i];
    Real lpPi, lpMu;
    if (bounded[i]) {
      const Real m = mean[i] > MAXM ? MAXM : (mean[i] < -MAXM ? -MAXM : mean[i]);   // getMean() :217-222
      const Real squash = std::tanh(act[i]), J = std::max(1 - squash * squash, FMIN);
      lpPi = -std::pow((act[i] - m) * invStd[i], 2) / 2 + std::log(invStd[i] / J) - LOG2PI_2;  // :240-249
      const Real bInv = 1 / bStd;
      lpMu = -std::pow((act[i] - bMean) * bInv, 2) / 2 + std::log(bInv / J) - LOG2PI_2;         // :274-279
    } else {
      lpPi = -std::pow((act[i] - mean[i]) * invStd[i], 2) / 2 + std::log(invStd[i]) - LOG2PI_2;   // :91-97
      const Real bInv = 1 / bStd;
      lpMu = -std::pow((act[i] - bMean) * bInv, 2) / 2 + std::log(bInv) - LOG2PI_2;
    }
    logW += lpPi - lpMu;                                                              // :648-653
    // KLdivergence with SMARTIES_OPPOSITE_KL (Settings/Bund.h:43): Dkl(pi||mu), raw mean (:286-298)
    const Real CmuCpi = std::pow(stdev[i] / bStd, 2);
    const Real sumDmeanC = std::pow((mean[i] - bMean) / bStd, 2);
    kl += (CmuCpi - 1 + sumDmeanC - std::log(CmuCpi)) / 2;
  }
  const Real RHO = std::exp(logW > 7 ? 7 : (logW < -7 ? -7 : logW));
  const bool far = isFarPolicy((Fval)RHO, (Fval)Cmax, (Fval)Cinv);
  const Real V = scaleNet2V(O[0]);
  // Gaussian_advantage::computeAdvantage (:76-81): coef (exp(-1/2 sum (a-m)^2 / L) - mixture ratio), L = L+ above the
  // mean, L- below; mean and variance are the policy's (getMean(): clipped for squashed components)
  Real Aval = 0, advCoef = 0, advOrig = 0, advRatio = 1;
  std::vector<Real> mat(2 * (size_t)dA, 1), pmean(dA);
  if (nAdv) {
    advCoef = spEval(O[1]);
    for (int i = 0; i < 2 * dA; ++i) mat[i] = spEval(O[2 + i]);
    Real quad = 0;
    for (int i = 0; i < dA; ++i) {
      pmean[i] = bounded[i] ? (mean[i] > MAXM ? MAXM : (mean[i] < -MAXM ? -MAXM : mean[i])) : mean[i];
      const int matind = act[i] > pmean[i] ? i : i + dA;                                   // diagInvMul (:118-128)
      quad += std::pow(act[i] - pmean[i], 2) / mat[matind];
      const Real S = stdev[i] * stdev[i];
      advRatio *= std::sqrt(mat[i] / (mat[i] + S)) / 2 + std::sqrt(mat[i + dA] / (mat[i + dA] + S)) / 2;   // coefMixRatio
    }
    advOrig = std::exp(-quad / 2);
    Aval = advCoef * (advOrig - advRatio);
  }
  const Real A_RET = Qret - V, dQ = A_RET - Aval;       // Zero_advantage: A == 0
  const Real Ver = std::min((Real)1, RHO) * dQ;
  const Real Aer = std::min(Cmax, RHO) * dQ;
  grad[0] = far ? 0 : Ver * beta * scaleVdiff(O[0]);     // RACER_train.cpp:51
  const Real coef = A_RET * std::min(Cmax, RHO);
  for (int i = 0; i < dA; ++i) {
    const Real bMean = mu[i], bStd = mu[dA + i];
    // gradKLdiv(mu, factor=-1) with OPPOSITE_KL (:318-334)
    const Real dMean = mean[i] - bMean;
    const Real invVarMu = 1 / std::pow(bStd, 2);
    const Real penalM = -1 * (dMean * invVarMu);
    const Real penalS = dPos[i] * -1 * ((invVarMu - std::pow(invStd[i], 2)) * stdev[i]);
    Real polM = 0, polS = 0;
    if (!far) {
      if (bounded[i]) {  // :300-316
        const Real dLogPdMean = (act[i] - mean[i]) * invStd[i] * invStd[i];
        const Real m = mean[i] > MAXM ? MAXM : (mean[i] < -MAXM ? -MAXM : mean[i]);
        const Real u = (act[i] - m) * invStd[i];
        const Real dLogPdStdv = (u * u - 1) * invStd[i];
        polS = dPos[i] * coef * dLogPdStdv;
        if (mean[i] >= MAXM && coef * dLogPdMean > 0) polM = 0;
        else if (mean[i] <= -MAXM && coef * dLogPdMean < 0) polM = 0;
        else polM = coef * dLogPdMean;
      } else {           // :149-156
        const Real u = (act[i] - mean[i]) * invStd[i];
        polM = coef * (u * invStd[i]);
        polS = dPos[i] * coef * ((u * u - 1) * invStd[i]);
      }
    }
    // Utilities::penalizeReFER (Utils/FunctionUtilities.h:221-228) + makeNetworkGrad (:727-738)
    grad[pM + i] = beta * polM + (1 - beta) * penalM;
    grad[pS + i] = beta * polS + (1 - beta) * penalS;
  }
  if (nAdv) {   // Gaussian_advantage::grad (:91-116) with Qer = far ? 0 : beta * Aer (RACER_train.cpp:56)
    const Real Qer = far ? 0 : beta * Aer, expect = -advRatio;
    grad[1] += advOrig + expect;
    for (int i = 0; i < dA; ++i) {
      const Real m = pmean[i], p1 = mat[i], p2 = mat[i + dA];
      grad[2 + i] = act[i] > m ? advOrig * advCoef * std::pow((act[i] - m) / p1, 2) / 2 : 0;
      grad[2 + dA + i] = act[i] < m ? advOrig * advCoef * std::pow((act[i] - m) / p2, 2) / 2 : 0;
      const Real S = stdev[i] * stdev[i];
      const Real F = 2 / (std::sqrt(p1 / (p1 + S)) + std::sqrt(p2 / (p2 + S)));
      const Real diff1 = S / std::sqrt(p1 * std::pow(p1 + S, 3)) / 4;
      const Real diff2 = S / std::sqrt(p2 * std::pow(p2 + S, 3)) / 4;
      grad[2 + i] += F * expect * advCoef * diff1;
      grad[2 + dA + i] += F * expect * advCoef * diff2;
    }
    for (int e = 0; e < nAdv; ++e) grad[1 + e] *= Qer * spDiff(O[1 + e]);                  // grad_matrix (:69-74)
  }
  *rho = RHO; *dkl = kl; *deltaQ = dQ; *isFar = far ? 1 : 0; *Vval = V; *Qval = Aval + V;
}
// RACER::Train with Discrete_policy (Math/Discrete_policy.h:17-208: SoftPlus-normalised probabilities) and
// Discrete_advantage (Math/Discrete_advantage.h:17-100).  Output layout [V | A x nOpt | logits x nOpt].
extern "C" void ol_head_discrete(int nOpt, const double* O, double actMsg, const double* mu, double Qret, double beta,
                                 double Cmax, double Cinv, double* grad, double* rho, double* dkl, double* deltaQ,
                                 int* isFar, double* Vval, double* Qval) {
  const int pA = 1, pP = 1 + nOpt;
  const int act = (int)std::floor(actMsg);                         // ActionInfo::actionMessage2label (StateAction.h:300-320)
  std::vector<Real> unnorm(nOpt), probs(nOpt);
  Real norm = 0;
  for (int j = 0; j < nOpt; ++j) { unnorm[j] = spEval(O[pP + j]); norm += unnorm[j]; }
  norm = std::max(norm, std::numeric_limits<Real>::epsilon());
  for (int j = 0; j < nOpt; ++j) probs[j] = unnorm[j] / norm;
  const Real RHO = probs[act] / mu[act];                           // importanceWeight (:84-91)
  Real kl = 0;
  for (int i = 0; i < nOpt; ++i) kl += probs[i] * std::log(probs[i] / mu[i]);   // KLDivergence (:126-130)
  const bool far = isFarPolicy((Fval)RHO, (Fval)Cmax, (Fval)Cinv);
  Real expA = 0;
  for (int j = 0; j < nOpt; ++j) expA += probs[j] * O[pA + j];
  const Real Aval = O[pA + act] - expA;                            // computeAdvantage (:64-70)
  const Real V = scaleNet2V(O[0]);
  const Real A_RET = Qret - V, dQ = A_RET - Aval;
  const Real Ver = std::min((Real)1, RHO) * dQ, Aer = std::min(Cmax, RHO) * dQ;
  grad[0] = far ? 0 : Ver * beta * scaleVdiff(O[0]);
  // penalG = KLDivGradient(mu, -1) (:152-162), polG = policyGradient(act, A_RET min(Cmax, rho)) (:136-144) or zeros
  std::vector<Real> penalG(nOpt, 0), polG(nOpt, 0);
  for (int j = 0; j < nOpt; ++j) {
    const Real tmp = -1 * (1 + std::log(probs[j] / mu[j])) / norm;
    for (int i = 0; i < nOpt; ++i) penalG[i] += tmp * ((i == j) - probs[j]);
  }
  for (int j = 0; j < nOpt; ++j) penalG[j] *= spDiff(O[pP + j]);
  if (!far) {
    const Real factor = A_RET * std::min(Cmax, RHO);
    polG[act] = factor / unnorm[act];
    for (int i = 0; i < nOpt; ++i) { polG[i] -= factor / norm; polG[i] *= spDiff(O[pP + i]); }
  }
  for (int j = 0; j < nOpt; ++j) grad[pP + j] = beta * polG[j] + (1 - beta) * penalG[j];   // penalizeReFER + makeNetworkGrad
  const Real Qer = far ? 0 : beta * Aer;
  for (int j = 0; j < nOpt; ++j) grad[pA + j] = Qer * ((j == act ? 1 : 0) - probs[j]);     // Discrete_advantage::grad (:51-58)
  *rho = RHO; *dkl = kl; *deltaQ = dQ; *isFar = far ? 1 : 0; *Vval = V; *Qval = Aval + V;
}

namespace {

// Episode::updateCumulative_atomic (ReplayMemory/Episode.h:112-129)
void epUpdateCumulative(Episode& EP, int t, Fval E, Fval D, Fval W, Fval C, Fval invC) {
  const Fval wasFarPol = EP.IMPW[t] > C || EP.IMPW[t] < invC;
  const Fval isFarPol = W > C || W < invC;
  const Fval invN = 1 / (Fval)EP.N;
  EP.avgKL += invN * (D - EP.DKL[t]);
  EP.fracFar += invN * (isFarPol - wasFarPol);
  EP.avgSqErr += invN * (E * E - EP.DQ[t] * EP.DQ[t]);
  EP.maxAbsErr = std::max(EP.maxAbsErr, std::fabs(E));
  EP.DQ[t] = E; EP.DKL[t] = D; EP.IMPW[t] = W;
}
// Episode::updateValues_atomic (Episode.h:131-145)
void epUpdateValues(Episode& EP, int t, Fval V, Fval Q) {
  const Fval oldQ = EP.ADV[t] + EP.V[t];
  EP.sumQ2 += Q * Q - oldQ * oldQ;
  EP.sumQ += Q - oldQ;
  EP.maxQ = std::max(EP.maxQ, Q);
  EP.minQ = std::min(EP.minQ, Q);
  EP.V[t] = V; EP.ADV[t] = Q - V;
}
// Episode::updateCumulative (ReplayMemory/Episode.cpp:213-242)
void epRecompute(Episode& EP, Fval C, Fval invC) {
  const int N = EP.ndata();
  const Fval invN = 1 / (Fval)N;
  int64_t nFarPol = 0;
  Fval sumE2 = 0, maxAE = -1e9, maxQ = -1e9, sumQ2 = 0, minQ = 1e9, sumQ1 = 0;
  for (int t = 0; t < N; ++t) {
    if (EP.IMPW[t] > C || EP.IMPW[t] < invC) ++nFarPol;
    sumE2 += EP.DQ[t] * EP.DQ[t];
    maxAE = std::max(maxAE, std::fabs(EP.DQ[t]));
    const Fval Q = EP.ADV[t] + EP.V[t];
    maxQ = std::max(maxQ, Q); minQ = std::min(minQ, Q);
    sumQ2 += Q * Q; sumQ1 += Q;
  }
  EP.fracFar = invN * nFarPol; EP.avgSqErr = invN * sumE2; EP.maxAbsErr = maxAE;
  EP.sumQ2 = sumQ2; EP.sumQ = sumQ1; EP.maxQ = maxQ; EP.minQ = minQ;
  Real tot = 0; for (int t = 0; t < EP.N; ++t) tot += EP.R[t];   // Utilities::sum over Real
  EP.totR = tot;
  Fval sk = 0; for (int t = 0; t < EP.N; ++t) sk += EP.DKL[t];    // Utilities::sum over Fval
  EP.avgKL = invN * sk;
}

// updateReturnEstimator (ReplayMemory/MemoryProcessing.cpp:23-44) with the estimator createReturnEstimator picks
// (:418-450): computeRetrace (:391-400), computeRetraceExplBonus (:402-408; its baseline is ReplayStats::maxAbsError as
// of the moment the estimator is created), computeGAE (:410-416).  Returns the sum of the squared changes (sumErr2).
Fval retraceEpisode(const ol_learner* h, Episode& EP) {
  const int kind = h->cfg.returnsEstimator;
  if (kind == HL_RET_NONE) return 0;
  const Fval gamma = h->cfg.gamma, lambda = h->cfg.lambda;
  const Fval coef = (1 - gamma), baseline = (Fval)h->stats.maxAbsError;
  if (!EP.term) EP.RET[EP.N - 1] = EP.V[EP.N - 1];
  Fval sumErr2 = 0;
  for (int t = EP.N - 2; t >= 0; --t) {
    const Fval oldEstimate = EP.RET[t];
    const Fval R = (Fval)((EP.R[t + 1] - h->rewMean) * h->rewScale);   // Episode::scaledReward<Fval> (:185-189)
    const Fval Q = EP.RET[t + 1], V = EP.V[t + 1], A = EP.ADV[t + 1];
    if (kind == HL_RET_GAE) EP.RET[t] = R + gamma * (V + lambda * (Q - V));
    else {
      const Fval w = EP.IMPW[t + 1] < 1 ? EP.IMPW[t + 1] : 1;          // clippedOffPolW (:191-195)
      const Fval ret = R + gamma * (V + lambda * w * (Q - A - V));
      if (kind == HL_RET_RETRACE_EXPLORE) { const Fval E = std::fabs(Q - A - V) - baseline; EP.RET[t] = coef * E + ret; }
      else EP.RET[t] = ret;
    }
    sumErr2 += std::pow(oldEstimate - EP.RET[t], 2);                   // (float, int) -> double arithmetic, stored as Fval
  }
  return sumErr2;
}

// MemoryProcessing::updateCounters (MemoryProcessing.cpp:46-92)
void updateCounters(ol_learner* h, bool /*bInit*/) {
  int64_t nFar = h->stats.nFarPolicySteps, nStored = h->nTransitions;
  h->seenEpsUpd = h->nSeenEps; h->seenStepsUpd = h->nSeenSteps;
  if (h->cfg.n_ranks > 1 && h->countersReduced) { nFar = h->nFarGlobal; nStored = h->nStoredGlobal; h->seenEpsUpd = h->seenEpsGlobal; h->seenStepsUpd = h->seenStepsGlobal; }
  h->countersReduced = false;
  const Real fracOffPol = nFar / (Real)std::max(nStored, (int64_t)1);
  const Real maxN = (Real)h->maxObsGlobal, BS = h->Bglobal;
  const Real nDataSize = std::max(maxN, (Real)nStored);
  const Real learnRefer = 0.1 * BS / nDataSize;
  const auto fixPointIter = [&](const Real val, const bool goTo0) {
    if (goTo0) return (1 - std::min(learnRefer, val)) * val;
    else return (1 - std::min(learnRefer, val)) * val + std::min(learnRefer, 1 - val);
  };
  h->beta = fixPointIter(h->beta, fracOffPol > h->cfg.penalTol);
  h->alpha = fixPointIter(h->alpha, std::fabs(h->cfg.penalTol - fracOffPol) < 1e-3);
}

// MemoryProcessing::updateRewardsStats (MemoryProcessing.cpp:94-185), split at the
// StateRewRdx all-reduce (:139-150): computeMoments = local sums, applyMoments = EMA update
void computeMoments(const ol_learner* h, std::vector<long double>& out) {
  const int dS = h->dS;
  long double count = 0, newRSum = 0, newRSqSum = 0;
  std::vector<long double> SSum(dS, 0), SSqSum(dS, 0);
  for (auto& ep : h->episodes) {
    const Episode& EP = *ep; const int N = EP.ndata();
    count += N;
    for (int j = 0; j < N; ++j) {
      const long double drk = EP.R[j + 1] - h->rewMean;
      newRSum += drk; newRSqSum += drk * drk;
      for (int k = 0; k < dS; ++k) {
        const long double dsk = EP.S[(size_t)j * dS + k] - h->stMean[k];  // float - float
        SSum[k] += dsk; SSqSum[k] += dsk * dsk;
      }
    }
  }
  out.assign(SSum.begin(), SSum.end());
  out.insert(out.end(), SSqSum.begin(), SSqSum.end());
  out.push_back(count); out.push_back(newRSum); out.push_back(newRSqSum);
}
void applyMoments(ol_learner* h, const std::vector<long double>& mom, bool bInit, Real rRateFac) {
  const int dS = h->dS;
  const Real learnR = (Real)h->cfg.learnrate / (1 + (Real)h->nGradSteps * h->cfg.epsAnneal);
  const Real annealLearnR = std::min((Real)1, rRateFac * learnR);
  const Real WS = bInit ? 1 : annealLearnR, WR = bInit ? 1 : annealLearnR;
  const long double count = mom[2 * dS];
  const auto updateStats = [](nnReal& mean, nnReal& stdev, nnReal& invstdev, const Real learnRate,
                              const long double Evar, const long double Evar2) {
    mean += learnRate * Evar;
    auto variance = Evar2 - Evar * Evar * (2 * learnRate - learnRate * learnRate);
    static constexpr long double EPS = FLT_EPSILON;
    variance = std::max(variance, EPS);
    stdev += learnRate * (std::sqrt(variance) - stdev);
    invstdev = 1 / stdev;
  };
  if (WR > 0) updateStats(h->rewMean, h->rewStd, h->rewScale, WR, mom[2 * dS + 1] / count, mom[2 * dS + 2] / count);
  if (WS > 0) for (int k = 0; k < dS; ++k)
    updateStats(h->stMean[k], h->stStd[k], h->stScale[k], WS, mom[k] / count, mom[dS + k] / count);
}

// MemoryProcessing::updateTrainingStatistics (MemoryProcessing.cpp:187-259), nThreads = 1
void updateTrainingStatistics(ol_learner* h) {
  const int64_t nGradSteps = h->nGradSteps + 1;
  const bool bRecompute = (nGradSteps % 1000) == 0;
  const Real C = h->cfg.clipImpWeight, E = h->cfg.epsAnneal;
  h->CmaxRet = 1 + C / (1 + (Real)nGradSteps * E);   // Utilities::annealRate (FunctionUtilities.h:69-72)
  h->CinvRet = 1 / h->CmaxRet;
  size_t nOffPol = 0;
  Fval maxAbsE = -1e9, maxQ = -1e9, minQ = 1e9;
  Real sumDKL = 0, sumE2 = 0, sumQ2 = 0, sumQ1 = 0, sumR = 0, sumERet = 0;
  const bool bNeedsReturnEst = h->cfg.returnsEstimator != HL_RET_NONE;
  size_t nRetUpdates = 0;
  for (auto& ep : h->episodes) {
    Episode& EP = *ep;
    if (bRecompute) {
      epRecompute(EP, (Fval)h->CmaxRet, (Fval)h->CinvRet);
      if (bNeedsReturnEst) { sumERet += retraceEpisode(h, EP); nRetUpdates += EP.N - 1; }
    }
    const Fval Nsteps = EP.N;
    maxAbsE = std::max(EP.maxAbsErr, maxAbsE);
    maxQ = std::max(EP.maxQ, maxQ); minQ = std::min(EP.minQ, minQ);
    sumDKL += Nsteps * EP.avgKL;
    nOffPol += Nsteps * EP.fracFar;   // size_t += float: float add, then truncation (:227)
    sumE2 += Nsteps * EP.avgSqErr;
    sumQ2 += EP.sumQ2; sumQ1 += EP.sumQ; sumR += EP.totR;
  }
  if (h->CmaxRet <= 1) nOffPol = 0;
  const int64_t nData = h->nTransitions; const size_t setSize = h->episodes.size();
  h->stats.nFarPolicySteps = (int64_t)nOffPol;
  const Real maxN = (Real)h->maxObsGlobal, BS = h->Bglobal;
  const Real learnRefer = 0.1 * BS / std::max(maxN, (Real)nData);
  h->stats.maxAbsError += learnRefer * (maxAbsE - h->stats.maxAbsError);
  h->stats.avgKLdivergence = sumDKL / nData;
  h->stats.avgSquaredErr = sumE2 / nData;
  h->stats.avgReturn = sumR / setSize;
  h->stats.avgQ = sumQ1 / nData;
  h->stats.maxQ = maxQ; h->stats.minQ = minQ;
  h->stats.stdevQ = sumQ2 / nData - h->stats.avgQ * h->stats.avgQ;
  h->stats.stdevQ = std::sqrt(std::max(h->stats.stdevQ, 1e-16));
  if (bNeedsReturnEst) {      // :250-258
    if (h->stats.countReturnsEstimateUpdates < 0) h->stats.countReturnsEstimateUpdates = 0;
    h->stats.countReturnsEstimateUpdates += (int64_t)nRetUpdates;
    h->stats.sumReturnsEstimateErrors += sumERet;
  } else { h->stats.countReturnsEstimateUpdates = -1; h->stats.sumReturnsEstimateErrors = 0; }
}

// MemoryProcessing::applyEpisodesRemovalAlgo, "oldest" filter (MemoryProcessing.cpp:261-275,327-351)
// MemoryProcessing::applyEpisodesRemovalAlgo + getERfilterAlgo (MemoryProcessing.cpp:261-351): sort the episodes so that the
// ones to delete are at the back, pop the back while the replay stays over budget without it
void applyEpisodesRemoval(ol_learner* h) {
  using EPtr = std::unique_ptr<Episode>;
  const int filter = h->cfg.ERoldSeqFilter;
  if (h->cfg.episode_order == HL_ORDER_REFERENCE) {      // the reference's own (non-stable) std::sort with its comparator
    if (filter == HL_ER_FARPOLFRAC) std::sort(h->episodes.begin(), h->episodes.end(), [](const EPtr& a, const EPtr& b) { return a->fracFar < b->fracFar; });
    else if (filter == HL_ER_MAXKLDIV) std::sort(h->episodes.begin(), h->episodes.end(), [](const EPtr& a, const EPtr& b) { return a->avgKL < b->avgKL; });
    else if (filter == HL_ER_MINERROR) std::sort(h->episodes.begin(), h->episodes.end(), [](const EPtr& a, const EPtr& b) { return a->avgSqErr > b->avgSqErr; });
    else std::sort(h->episodes.begin(), h->episodes.end(), [](const EPtr& a, const EPtr& b) { return a->ID > b->ID; });
    while (!h->episodes.empty() && h->nTransitions - (int64_t)h->episodes.back()->N > h->maxObsLocal) {
      h->nTransitions -= h->episodes.back()->ndata();
      h->episodes.pop_back();
    }
    return;
  }
  // product semantics (HL_ORDER_STABLE): the sampling order stays newest episode first (IDs are non-decreasing in insertion
  // order, so for "oldest" this is one of the orders the reference's comparator admits); the victim is the episode the
  // comparator puts last, the older one among equal keys
  std::sort(h->episodes.begin(), h->episodes.end(), [](const EPtr& a, const EPtr& b) { return a->seq > b->seq; });
  while (!h->episodes.empty()) {
    size_t v = h->episodes.size() - 1;
    if (filter != HL_ER_OLDEST) {
      auto key = [&](const Episode& e) -> Fval { return filter == HL_ER_FARPOLFRAC ? e.fracFar : (filter == HL_ER_MAXKLDIV ? e.avgKL : -e.avgSqErr); };
      for (size_t i = h->episodes.size(); i-- > 0;) if (key(*h->episodes[i]) > key(*h->episodes[v])) v = i;      // (scan from the oldest: ties keep it)
    }
    if (h->nTransitions - (int64_t)h->episodes[v]->N <= h->maxObsLocal) break;
    h->nTransitions -= h->episodes[v]->ndata();
    h->episodes.erase(h->episodes.begin() + (long)v);
  }
}

// Sample_uniform::sample + Sampling::IDtoSeqStep (ReplayMemory/Sampling.cpp:26-47,82-96)
void sampleUniform(ol_learner* h, std::vector<int64_t>& flat) {
  const int B = h->B; const uint64_t nData = (uint64_t)h->nTransitions;
  flat.resize(B);
  size_t it = 0;
  while (it != (size_t)B) {
    for (size_t i = it; i < (size_t)B; ++i) flat[i] = (int64_t)uniformIndex(h->gen, nData);
    std::sort(flat.begin(), flat.end());
    it = std::unique(flat.begin(), flat.end()) - flat.begin();
  }
}
// ---- prioritised samplers (ReplayMemory/Sampling.cpp:101-296) over libstdc++'s std::discrete_distribution<Uint>, restated:
//   param_type::_M_initialize   sum = accumulate(p, 0.0); p /= sum; cp = partial_sum(p); cp.back() = 1   (all sequential, double)
//   operator()                  lower_bound(cp, generate_canonical<double, 53>(urng)) - cp.begin()
//   generate_canonical<double,53> over mt19937: (w0 + w1 * 2^32) / 2^64 from two words, nextafter(1, 0) if it rounds to 1
struct DiscreteDist {
  std::vector<double> cp;
  void init(const std::vector<float>& p) {
    cp.clear();
    if (p.size() < 2) return;
    std::vector<double> q(p.begin(), p.end());
    double sum = 0.0; for (double v : q) sum += v;
    for (double& v : q) v /= sum;
    cp.resize(q.size()); double acc = 0; bool first = true;
    for (size_t i = 0; i < q.size(); ++i) { acc = first ? q[i] : acc + q[i]; first = false; cp[i] = acc; }
    cp.back() = 1.0;
  }
  size_t draw(MT19937& g) const {
    if (cp.empty()) return 0;
    double sum = 0, tmp = 1;
    for (int k = 0; k < 2; ++k) { sum += (double)g.next() * tmp; tmp *= 4294967296.0; }
    double r = sum / tmp; if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return (size_t)(std::lower_bound(cp.begin(), cp.end(), r) - cp.begin());
  }
};
void perPrepare(ol_learner* h, DiscreteDist& dist) {
  const float EPS = std::numeric_limits<float>::epsilon();
  const int algo = h->cfg.dataSamplingAlgo;
  std::vector<float> probs;
  if (algo == HL_SAMPLE_PERSEQ) {          // Sample_impSeq::prepare (:234-258)
    for (const auto& ep : h->episodes) probs.push_back(std::sqrt(std::sqrt(ep->avgSqErr + EPS)) * (float)(size_t)ep->ndata());
  } else if (algo == HL_SAMPLE_PERERR) {   // TSample_impErr::prepare (:173-205)
    for (const auto& ep : h->episodes) for (int j = 0; j < ep->ndata(); ++j) { const float d2 = ep->DQ[j] * ep->DQ[j]; probs.push_back(std::sqrt(std::sqrt(d2 + EPS))); }
  } else {                                  // TSample_impRank::prepare (:102-149)
    using Tup = std::tuple<float, unsigned, unsigned>;
    std::vector<Tup> errors; std::vector<size_t> prefixes; size_t prefix = 0;
    for (size_t i = 0; i < h->episodes.size(); ++i) {
      prefixes.push_back(prefix); prefix += (size_t)h->episodes[i]->ndata();
      for (int j = 0; j < h->episodes[i]->ndata(); ++j) errors.emplace_back(h->episodes[i]->DQ[j] * h->episodes[i]->DQ[j], (unsigned)i, (unsigned)j);
    }
    const auto before = [](const Tup& a, const Tup& b) { return std::get<0>(a) > std::get<0>(b); };
    if (h->cfg.episode_order == HL_ORDER_REFERENCE) std::sort(errors.begin(), errors.end(), before);      // the reference's non-stable sort
    else std::stable_sort(errors.begin(), errors.end(), before);                                         // product: equal errors in storage order
    probs.assign(errors.size(), 1.f);
    for (unsigned i = 0; i < errors.size(); ++i) {
      const float P = std::get<0>(errors[i]) > 0 ? 1 / std::sqrt(std::sqrt(i + 1)) : 1;                   // (integer argument: double square roots, :141)
      probs[prefixes[std::get<1>(errors[i])] + std::get<2>(errors[i])] = P;
    }
  }
  dist.init(probs);
}
// TSample_impRank / TSample_impErr::sample (:151-170, 207-230) and Sample_impSeq::sample, transitions (:270-294)
void samplePER(ol_learner* h, std::vector<int64_t>& flat) {
  DiscreteDist dist; perPrepare(h, dist);
  const int B = h->B; flat.resize(B);
  size_t it = 0;
  if (h->cfg.dataSamplingAlgo == HL_SAMPLE_PERSEQ) {
    std::vector<int64_t> prefix; int64_t acc = 0;
    for (const auto& ep : h->episodes) { prefix.push_back(acc); acc += ep->ndata(); }
    while (it != (size_t)B) {
      for (size_t i = it; i < (size_t)B; ++i) {
        const size_t s = dist.draw(h->gen);
        const size_t t = (size_t)(uniformFloat(h->gen, 0.f, 1.f) * (float)(size_t)h->episodes[s]->ndata());
        flat[i] = prefix[s] + (int64_t)t;        // (sorting / uniquing (episode, step) pairs == sorting / uniquing these)
      }
      std::sort(flat.begin(), flat.end());
      it = std::unique(flat.begin(), flat.end()) - flat.begin();
    }
    return;
  }
  while (it != (size_t)B) {
    for (size_t i = it; i < (size_t)B; ++i) flat[i] = (int64_t)dist.draw(h->gen);
    std::sort(flat.begin(), flat.end());
    it = std::unique(flat.begin(), flat.end()) - flat.begin();
  }
}
void idToSeqStep(ol_learner* h, const std::vector<int64_t>& flat, std::vector<int64_t>& seq,
                 std::vector<int64_t>& obs) {
  const size_t B = flat.size(); seq.assign(B, 0); obs.assign(B, 0);
  size_t i = 0; int64_t prefix = 0;
  for (size_t k = 0; k < h->episodes.size() && i < B; ++k) {
    const int64_t nsteps = h->episodes[k]->ndata();
    while (i < B && flat[i] < prefix + nsteps) { obs[i] = flat[i] - prefix; seq[i] = (int64_t)k; ++i; }
    prefix += nsteps;
  }
}

// Adam::step + AdamOptimizer::apply_update (Network/Optimizer.cpp:61-108,122-160)
void adamApply(ol_learner* h) {
  const Real factor = 1.0 / h->Bglobal;
  // Optimizer.h:52-53: nnReal eta = eta_init; annealRate<nnReal>(eta, nStep, epsAnneal)
  const nnReal eta0 = (nnReal)h->cfg.learnrate;
  const nnReal _eta = (nnReal)(eta0 / (1 + (nnReal)h->nStep * h->cfg.epsAnneal));
  (void)h->gen.next();  // Saru seed draw of thread 0 (:139): the other threads draw from generators of their own
  const nnReal betat1 = (nnReal)h->beta_t_1, betat2 = (nnReal)h->beta_t_2;
  const nnReal eta = _eta * std::sqrt(1 - betat2) / (1 - betat1);
  const nnReal B1 = (nnReal)0.9, B2 = (nnReal)0.999, lambda = (nnReal)h->cfg.nnLambda, fac = (nnReal)factor;
  nnReal* Wp = h->W.data(); nnReal* M1 = h->M1.data(); nnReal* M2 = h->M2.data(); nnReal* Gp = h->G.data();
  for (int64_t i = 0; i < h->nParams; ++i) {
    const nnReal penal = -Wp[i] * lambda;
    const nnReal DW = fac * Gp[i];
    M1[i] = B1 * M1[i] + (1 - B1) * DW;
    M2[i] = B2 * M2[i] + (1 - B2) * DW * DW;
    const nnReal numer = B1 * M1[i] + (1 - B1) * DW;   // SMARTIES_NESTEROV_ADAM
    M2[i] = M2[i] < M1[i] * M1[i] ? M1[i] * M1[i] : M2[i];  // SMARTIES_SAFE_ADAM
    const nnReal ret = numer / (nnEPS + std::sqrt(M2[i]));
    Wp[i] += eta * (ret + penal);                       // SMARTIES_ADAMW
  }
  std::fill(h->G.begin(), h->G.end(), 0);
  h->beta_t_1 *= 0.9; if (h->beta_t_1 < nnEPS) h->beta_t_1 = 0;
  h->beta_t_2 *= 0.999; if (h->beta_t_2 < nnEPS) h->beta_t_2 = 0;
}

}  // namespace

// ---------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------
extern "C" {

int ol_create(const hl_config* cfg, ol_learner** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(hl_config)) return HL_ERR_BAD_ARG;
  if (cfg->dimS <= 0 || cfg->dimA <= 0 || cfg->dimA > HL_MAX_DIMA || cfg->n_hidden < 1 ||
      cfg->n_hidden > HL_MAX_HIDDEN || cfg->batchSize <= 0 || cfg->n_ranks < 1) return HL_ERR_BAD_ARG;
  if (cfg->adv_kind != HL_ADV_ZERO && cfg->adv_kind != HL_ADV_GAUSSIAN && cfg->adv_kind != HL_ADV_DISCRETE) return HL_ERR_UNSUPPORTED;
  if (cfg->adv_kind == HL_ADV_DISCRETE && (cfg->dimA != 1 || cfg->n_options < 2 || cfg->n_options > 64)) return HL_ERR_BAD_ARG;
  if (cfg->nnFunc < HL_FUNC_LINEAR || cfg->nnFunc > HL_FUNC_EXP) return HL_ERR_UNSUPPORTED;
  if (cfg->nnOutputFunc < HL_FUNC_LINEAR || cfg->nnOutputFunc > HL_FUNC_EXP) return HL_ERR_UNSUPPORTED;
  if (cfg->returnsEstimator < HL_RET_RETRACE || cfg->returnsEstimator > HL_RET_NONE) return HL_ERR_BAD_ARG;
  if (cfg->nn_type < HL_NN_FFNN || cfg->nn_type > HL_NN_RNN) return HL_ERR_UNSUPPORTED;
  if (cfg->n_encoder < 0 || cfg->n_encoder + cfg->n_hidden > HL_MAX_HIDDEN) return HL_ERR_BAD_ARG;
  if (cfg->ERoldSeqFilter < HL_ER_OLDEST || cfg->ERoldSeqFilter > HL_ER_MINERROR) return HL_ERR_BAD_ARG;
  if (cfg->dataSamplingAlgo < HL_SAMPLE_UNIFORM || cfg->dataSamplingAlgo > HL_SAMPLE_PERSEQ) return HL_ERR_BAD_ARG;
  if (cfg->nAppendedObs < 0 || cfg->n_conv < 0 || cfg->n_conv > HL_MAX_CONV) return HL_ERR_BAD_ARG;
  for (int j = 0; j < cfg->n_conv; ++j) {   // each layer takes the previous one's image; the first one the stacked input or its first part
    const hl_conv2d& d = cfg->conv[j];
    const int inSize = d.inpFeatures * d.inpY * d.inpX;
    const int prev = j == 0 ? cfg->dimS * (1 + cfg->nAppendedObs) : cfg->conv[j - 1].outFeatures * cfg->conv[j - 1].outY * cfg->conv[j - 1].outX;
    if ((j == 0 ? inSize > prev : inSize != prev) || d.outFeatures < 1 || d.outY < 1 || d.outX < 1 || d.stridex < 1 || d.stridey < 1) return HL_ERR_BAD_ARG;
    if (d.outY != (d.inpY - d.filtery + 2 * d.paddiny) / d.stridey + 1 || d.outX != (d.inpX - d.filterx + 2 * d.paddinx) / d.stridex + 1) return HL_ERR_BAD_ARG;
  }
  auto* h = new ol_learner(); h->cfg = *cfg;
  if (cfg->n_encoder > 0) {     // createEncoder: the encoder layers are the first hidden layers of the one network (Learner_approximator.cpp:149-166)
    int n = 0;
    for (int j = 0; j < cfg->n_encoder; ++j) if (cfg->encoder[j] > 0) h->cfg.hidden[n++] = cfg->encoder[j];
    h->nEncLayers = n;
    for (int j = 0; j < cfg->n_hidden; ++j) h->cfg.hidden[n++] = cfg->hidden[j];
    h->cfg.n_hidden = n; h->cfg.n_encoder = 0;
  }
  if (cfg->encoder_rnn && cfg->nn_type != HL_NN_MGU) { delete h; return HL_ERR_BAD_ARG; }
  h->dS = cfg->dimS; h->dA = cfg->dimA;
  // HyperParameters::defineDistributedLearning (Settings/HyperParameters.cpp:177-205)
  const Real nL = cfg->n_ranks;
  h->Bglobal = cfg->batchSize > 1 ? (int)(std::ceil(cfg->batchSize / nL) * nL) : cfg->batchSize;
  h->B = cfg->batchSize > 1 ? h->Bglobal / cfg->n_ranks : h->Bglobal;
  h->maxObsGlobal = (int64_t)(std::ceil(cfg->maxTotObsNum / nL) * nL);
  h->maxObsLocal = h->maxObsGlobal / cfg->n_ranks;
  int64_t minObs = cfg->minTotObsNum <= 0 ? cfg->maxTotObsNum : cfg->minTotObsNum;
  minObs = std::min(minObs, cfg->maxTotObsNum);
  minObs = (int64_t)(std::ceil(minObs / nL) * nL);
  h->minObsLocal = minObs / cfg->n_ranks;
  buildNet(h);
  h->gen.seed((uint32_t)(cfg->randSeed + (uint64_t)cfg->rank));   // ExecutionInfo.cpp:387,391
  for (int t = 1; t < cfg->ref_threads; ++t) (void)h->gen.next();   // ... :392-393: T - 1 generators of the other threads seeded from it
  h->stMean.assign(h->dS, 0); h->stStd.assign(h->dS, 1); h->stScale.assign(h->dS, 1);
  h->beta = cfg->clipImpWeight <= 0 ? 1 : 1e-4;                     // MemoryBuffer.h:41-44
  h->CmaxRet = 1 + cfg->clipImpWeight; h->CinvRet = 1 / cfg->clipImpWeight;
  *out = h; return HL_OK;
}
int ol_destroy(ol_learner* h) { delete h; return HL_OK; }
const char* ol_last_error(const ol_learner* h) { return h ? h->err.c_str() : "null handle"; }
int64_t ol_num_params(const ol_learner* h) { return h ? h->nParams : -1; }
int32_t ol_num_outputs(const ol_learner* h) { return h ? h->nOut : -1; }
int32_t ol_num_layers(const ol_learner* h) { return h ? (int32_t)h->layers.size() : -1; }
int ol_param_layout(const ol_learner* h, int64_t* indW, int64_t* nW, int64_t* indB, int64_t* nB) {
  if (!h) return HL_ERR_BAD_ARG;
  for (size_t l = 0; l < h->layers.size(); ++l) {
    if (indW) indW[l] = h->layers[l].indW; if (nW) nW[l] = h->layers[l].nW;
    if (indB) indB[l] = h->layers[l].indB; if (nB) nB[l] = h->layers[l].nB;
  }
  return HL_OK;
}
int ol_init_weights(ol_learner* h) { if (!h) return HL_ERR_BAD_ARG; initWeights(h); return HL_OK; }
int ol_set_params(ol_learner* h, const float* w, const float* m1, const float* m2) {
  if (!h) return HL_ERR_BAD_ARG;
  if (w) std::copy(w, w + h->nParams, h->W.begin());
  if (m1) std::copy(m1, m1 + h->nParams, h->M1.begin());
  if (m2) std::copy(m2, m2 + h->nParams, h->M2.begin());
  return HL_OK;
}
int ol_get_params(ol_learner* h, float* w, float* m1, float* m2) {
  if (!h) return HL_ERR_BAD_ARG;
  if (w) std::copy(h->W.begin(), h->W.end(), w);
  if (m1) std::copy(h->M1.begin(), h->M1.end(), m1);
  if (m2) std::copy(h->M2.begin(), h->M2.end(), m2);
  return HL_OK;
}
int ol_set_rng_state(ol_learner* h, const uint32_t s[625]) {
  if (!h || !s) return HL_ERR_BAD_ARG;
  std::memcpy(h->gen.x, s, 624 * 4); h->gen.p = s[624]; return HL_OK;
}
int ol_get_rng_state(ol_learner* h, uint32_t s[625]) {
  if (!h || !s) return HL_ERR_BAD_ARG;
  std::memcpy(s, h->gen.x, 624 * 4); s[624] = h->gen.p; return HL_OK;
}

// MemoryBuffer::addEpisodeToTrainingSet + Episode::finalize + pushBackEpisode
// (MemoryBuffer.cpp:131-170, 479-520; Episode.cpp:244-267, 269-274)
int ol_append_episode(ol_learner* h, int32_t N, const float* states, const double* actions,
                      const double* mu, const double* rewards, const float* values,
                      const float* advantages, int32_t terminated, int64_t tag) {
  if (!h || !states || !actions || !mu || !rewards || !values) return HL_ERR_BAD_ARG;
  if (N < 2) return fail(h, HL_ERR_BAD_ARG, "Episode must at least have s0 and sT");
  auto EP = std::make_unique<Episode>();
  const int dS = h->dS, dA = h->dA;
  EP->N = N; EP->term = terminated != 0; EP->tag = tag;
  EP->S.assign(states, states + (size_t)N * dS);
  EP->A.assign(actions, actions + (size_t)N * dA);
  EP->MU.assign(mu, mu + (size_t)N * h->polDim);
  EP->R.assign(rewards, rewards + N);
  EP->V.assign(values, values + N);
  if (advantages) EP->ADV.assign(advantages, advantages + N); else EP->ADV.assign(N, 0);
  for (int t = 1; t < N; ++t) EP->totR += rewards[t];          // MemoryBuffer.cpp:95-96
  h->nSeenSteps += N - 2;                                        // storeAction, t = 1..N-2 (:110)
  EP->DQ.assign(N, 0); EP->DKL.assign(N, 0); EP->IMPW.assign(N, 1); EP->IMPW[N - 1] = 0;
  EP->RET.assign(N, 0);
  retraceEpisode(h, *EP);                                        // computeReturnEstimator (:138)
  // pushBackEpisode: placeholder error (:486) and ID = max(nLocTimeStepsTrain(), 0) (:484, :515),
  // evaluated BEFORE the final increaseLocalSeenSteps of addEpisodeToTrainingSet (:167)
  const Real EPS = FLT_EPSILON;
  const Fval maxError = (Fval)std::sqrt(std::max(EPS, h->stats.avgSquaredErr));
  std::fill(EP->DQ.begin(), EP->DQ.end(), maxError);
  EP->avgSqErr = maxError * maxError; EP->maxAbsErr = maxError;
  const int64_t locTrain = h->nGatheredB4Startup == INT64_MAX ? -1 : h->nSeenSteps - h->nGatheredB4Startup;
  EP->ID = std::max(locTrain, (int64_t)0);
  if (!h->episodeLog.empty()) {    // MemoryBuffer.cpp:492-503: nGradSteps, time stamp, agent, steps, total reward
    if (FILE* f = fopen(h->episodeLog.c_str(), "a")) { fprintf(f, "%ld %ld %d %u %f\n", (long)h->nGradSteps, (long)EP->ID, 0, (unsigned)N, EP->totR); fclose(f); }
  }
  h->nSeenSteps += 1;                                            // :167
  EP->seq = h->nSeenEps;
  h->nTransitions += EP->ndata();
  if (h->cfg.episode_order == HL_ORDER_REFERENCE) h->episodes.push_back(std::move(EP));
  else h->episodes.insert(h->episodes.begin(), std::move(EP));   // newest first
  h->nSeenEps += 1;
  return HL_OK;
}

int ol_get_scaling(ol_learner* h, float* m, float* s, float* r3) {
  if (!h) return HL_ERR_BAD_ARG;
  if (m) std::copy(h->stMean.begin(), h->stMean.end(), m);
  if (s) std::copy(h->stScale.begin(), h->stScale.end(), s);
  if (r3) { r3[0] = h->rewMean; r3[1] = h->rewScale; r3[2] = h->rewStd; }
  return HL_OK;
}
int ol_set_scaling(ol_learner* h, const float* m, const float* s, const float* r3) {
  if (!h) return HL_ERR_BAD_ARG;
  if (m) std::copy(m, m + h->dS, h->stMean.begin());
  if (s) { std::copy(s, s + h->dS, h->stScale.begin()); for (int k = 0; k < h->dS; ++k) h->stStd[k] = 1 / s[k]; }
  if (r3) { h->rewMean = r3[0]; h->rewScale = r3[1]; h->rewStd = r3[2]; }
  return HL_OK;
}
int ol_get_episode_info(ol_learner* h, int64_t pos, int64_t* tag, int32_t* nsteps, int32_t* term) {
  if (!h || pos < 0 || pos >= (int64_t)h->episodes.size()) return HL_ERR_BAD_ARG;
  const Episode& EP = *h->episodes[pos];
  if (tag) *tag = EP.tag; if (nsteps) *nsteps = EP.N; if (term) *term = EP.term;
  return HL_OK;
}
int ol_get_episode_stats(ol_learner* h, int64_t pos, float* dst) {
  if (!h || !dst || pos < 0 || pos >= (int64_t)h->episodes.size()) return HL_ERR_BAD_ARG;
  const Episode& EP = *h->episodes[pos];
  const float v[9] = {EP.totR, EP.avgKL, EP.fracFar, EP.avgSqErr, EP.maxAbsErr, EP.sumQ2, EP.sumQ, EP.maxQ, EP.minQ};
  std::copy(v, v + 9, dst);
  return HL_OK;
}
int ol_get_episode_field(ol_learner* h, int64_t pos, int32_t field, float* dst, int32_t cap) {
  if (!h || !dst || pos < 0 || pos >= (int64_t)h->episodes.size()) return HL_ERR_BAD_ARG;
  const Episode& EP = *h->episodes[pos];
  if (cap < EP.N) return HL_ERR_BAD_ARG;
  const std::vector<float>* v = nullptr;
  switch (field) {
    case HL_EP_RETURN: v = &EP.RET; break; case HL_EP_VALUE: v = &EP.V; break;
    case HL_EP_ADVANTAGE: v = &EP.ADV; break; case HL_EP_IMPW: v = &EP.IMPW; break;
    case HL_EP_DKL: v = &EP.DKL; break; case HL_EP_DELTAQ: v = &EP.DQ; break;
    default: return HL_ERR_BAD_ARG;
  }
  std::copy(v->begin(), v->end(), dst); return HL_OK;
}

// Learner::initializeLearner (Learners/Learner.cpp:47-72), split at its two accurate reductions (updateCounters(true), updateRewardsStats(true):
// DelayedReductor::get(true) waits, so with several learners every one starts from the GLOBAL counters and reward / state moments)
int ol_initialize_begin(ol_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  if (h->episodes.empty()) return fail(h, HL_ERR_TOO_FEW_DATA, "empty replay");
  computeMoments(h, h->moments);
  h->momentsPending = true; h->initPending = true;
  return HL_OK;
}
int ol_initialize_end(ol_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  if (!h->initPending) return fail(h, HL_ERR_STATE, "initialize_end without initialize_begin");
  updateCounters(h, true);
  applyMoments(h, h->moments, true, 1);
  h->momentsPending = false; h->initPending = false;
  h->nGatheredB4Startup = h->minObsLocal;
  for (auto& ep : h->episodes) retraceEpisode(h, *ep);   // rescaleAllReturnEstimator (:460-481)
  h->initialized = true; return HL_OK;
}
int ol_initialize(ol_learner* h) {
  const int rc = ol_initialize_begin(h);
  return rc ? rc : ol_initialize_end(h);
}

// Learner_approximator::spawnTrainTasks (Learner_approximator.cpp:36-92), nThreads = 1
// Episode::standardizedState (Episode.h:172-183): the observed state of step t followed by the nAppendedObs previous ones,
// each (s - mean) * scale.  The reference indexes the appended steps with std::max((Uint) samp - j, (Uint) 0), which wraps
// for samp < j (an out-of-bounds read there); the evident intent -- steps before the first repeat the first -- is what
// both the library and this restatement do, and the fixtures of the compiled reference only sample t >= nAppendedObs.
static void standardizedState(const ol_learner* h, const Episode& EP, int t, nnReal* ret) {
  const int dS = h->dS, nApp = h->cfg.nAppendedObs;
  for (int j = 0, k = 0; j <= nApp; ++j) {
    const int tt = std::max(t - j, 0);
    for (int i = 0; i < dS; ++i, ++k) ret[k] = (EP.S[(size_t)tt * dS + i] - h->stMean[i]) * h->stScale[i];
  }
}

int ol_step_begin(ol_learner* h, const int64_t* flat_in) {
  if (!h) return HL_ERR_BAD_ARG;
  if (!h->initialized) return fail(h, HL_ERR_STATE, "step before initialize");
  if (h->inStep) return fail(h, HL_ERR_STATE, "step_begin twice");
  if (h->minObsLocal < h->cfg.batchSize && false) return HL_ERR_TOO_FEW_DATA;
  if (h->nTransitions < h->B) return fail(h, HL_ERR_TOO_FEW_DATA, "Parameter minTotObsNum is too low for given problem");
  const int B = h->B, dS = h->dS * (1 + h->cfg.nAppendedObs), dA = h->dA, nOut = h->nOut;      // dS: network input size
  if (flat_in) h->bFlat.assign(flat_in, flat_in + B);
  else if (h->cfg.dataSamplingAlgo == HL_SAMPLE_UNIFORM) sampleUniform(h, h->bFlat);
  else samplePER(h, h->bFlat);
  idToSeqStep(h, h->bFlat, h->bEp, h->bT);
  h->bTag.resize(B);
  if (h->tap) { h->tState.assign((size_t)B * dS, 0); h->tO.assign((size_t)B * nOut, 0); h->tG.assign((size_t)B * nOut, 0);
    h->tRho.assign(B, 0); h->tDkl.assign(B, 0); h->tDq.assign(B, 0); h->tFar.assign(B, 0); }
  std::vector<nnReal> inp(dS); std::vector<Real> O(nOut), On(nOut), grad(nOut);
  const size_t outParam = h->layers.size() - 1, outDense = h->nOpt ? outParam : outParam - 1;   // discrete: no sigma layer
  h->gsSum.assign(nOut, 0); h->gsSq.assign(nOut, 0);
  for (int b = 0; b < B; ++b) {
    Episode& EP = *h->episodes[h->bEp[b]]; const int t = (int)h->bT[b];
    h->bTag[b] = EP.tag;
    // MemoryBuffer::sampleMinibatch gather: Episode::standardizedState (Episode.h:172-183)
    standardizedState(h, EP, t, inp.data());
    if (h->tap) std::copy(inp.begin(), inp.end(), h->tState.begin() + (size_t)b * dS);
    // recurrent nets: the window of MemoryBuffer::sampleMinibatch (:391-402), min(nnBPTTseq, t) steps before t; every
    // step is forwarded with the previous one as recurrent input (Approximator::forward, Approximator.h:116-173)
    const bool recurrent = h->cfg.nn_type != HL_NN_FFNN;
    std::vector<Act> series; int T = 0;
    if (recurrent) {
      const int nBPTT = h->cfg.nnBPTTseq > 0 ? h->cfg.nnBPTTseq : 16;
      const int beg = t - std::min(nBPTT, t); T = t - beg;
      series.resize((size_t)T + 2);
      for (auto& a : series) { a.X = h->X; a.Y = h->Y; a.E = h->E; for (auto& e : a.E) std::fill(e.begin(), e.end(), 0); }
      std::vector<nnReal> in2(dS);
      for (int k = 0; k <= T; ++k) {
        standardizedState(h, EP, beg + k, in2.data());
        forwardNet(h, in2.data(), series[k].X, series[k].Y, k ? &series[k - 1].Y : nullptr);
      }
      getOutput(h, series[T].Y, O.data());
    } else { forwardNet(h, inp.data(), h->X, h->Y); getOutput(h, h->Y, O.data()); }
    if (EP.isTruncated(t + 1)) {   // RACER_train.cpp:23-27
      std::vector<nnReal> inpn(dS);
      standardizedState(h, EP, t + 1, inpn.data());
      if (recurrent) { forwardNet(h, inpn.data(), series[T + 1].X, series[T + 1].Y, &series[T].Y); getOutput(h, series[T + 1].Y, On.data()); }
      else { forwardNet(h, inpn.data(), h->Xn, h->Yn); getOutput(h, h->Yn, On.data()); }
      const Fval Vn = (Fval)scaleNet2V(On[0]);
      epUpdateValues(EP, t + 1, Vn, Vn);
    }
    Real rho, dkl, dq, V, Q; int far;
    const int nAdv = h->nAdv, nDn = h->nOpt ? 1 + 2 * h->nOpt : 1 + nAdv + dA;
    if (h->nOpt) ol_head_discrete(h->nOpt, O.data(), EP.A[(size_t)t * dA], &EP.MU[(size_t)t * h->polDim], (Real)EP.RET[t], h->beta,
                                  h->CmaxRet, h->CinvRet, grad.data(), &rho, &dkl, &dq, &far, &V, &Q);
    else ol_head_racer(dA, nAdv, h->cfg.bounded, O.data(), &EP.A[(size_t)t * dA], &EP.MU[(size_t)t * h->polDim],
                  (Real)EP.RET[t], h->beta, h->CmaxRet, h->CinvRet, grad.data(), &rho, &dkl, &dq, &far, &V, &Q);
    for (int o = 0; o < nOut; ++o) { h->gsSum[o] += grad[o]; h->gsSq[o] += grad[o] * grad[o]; }   // StatsTracker::track_vector (Approximator.h:197)
    // Approximator::setGradient -> Activation::addOutputDelta (Approximator.h:190-204, Activation.h:108-117)
    auto& Eout = recurrent ? series[T].E : h->E;
    for (auto& e : h->E) std::fill(e.begin(), e.end(), 0);
    for (int o = 0; o < nDn; ++o) Eout[outDense][o] += grad[o];
    if (!h->nOpt) for (int o = 0; o < dA; ++o) Eout[outParam][o] += grad[nDn + o];
    if (h->tap) { for (int o = 0; o < nOut; ++o) { h->tO[(size_t)b * nOut + o] = O[o]; }
      for (int o = 0; o < nDn; ++o) h->tG[(size_t)b * nOut + o] = Eout[outDense][o];
      if (!h->nOpt) for (int o = 0; o < dA; ++o) h->tG[(size_t)b * nOut + nDn + o] = Eout[outParam][o];
      h->tRho[b] = rho; h->tDkl[b] = dkl; h->tFar[b] = (uint8_t)far; }
    // MiniBatch::setMseDklImpw / setValues (RACER_train.cpp:59-60; MiniBatch.h:161-175)
    epUpdateCumulative(EP, t, (Fval)dq, (Fval)dkl, (Fval)rho, (Fval)h->CmaxRet, (Fval)h->CinvRet);
    epUpdateValues(EP, t, (Fval)V, (Fval)Q);
    if (h->tap) h->tDq[b] = EP.DQ[t];
    if (recurrent) backwardSeries(h, series, T); else backwardNet(h);
  }
  if (h->tap) h->tGradSum = h->G;
  {   // StatsTracker::reduce_stats (StatsTracker.cpp:100-107) as called by Learner_approximator.cpp:89 with iter = nGradSteps
    h->gsMean.resize(nOut); h->gsRms.resize(nOut);
    const long double cnt = std::max((long double)2.2e-16, (long double)B);
    for (int o = 0; o < nOut; ++o) { h->gsMean[o] = (Real)(h->gsSum[o] / cnt); h->gsRms[o] = (Real)std::sqrt((Real)(h->gsSq[o] / cnt)); }
    if (!h->logBase.empty() && h->nGradSteps % 1000 == 0 && h->cfg.rank == 0) {
      FILE* f = fopen((h->logBase + "_net_outGrad_stats.raw").c_str(), h->gsCalls ? "ab" : "wb");
      if (!f) return fail(h, HL_ERR_IO, "unable to open " + h->logBase + "_net_outGrad_stats.raw");
      if (!h->gsCalls) { const float hd = nOut + .1; fwrite(&hd, sizeof(float), 1, f); }
      std::vector<float> v(2 * (size_t)nOut);
      for (int o = 0; o < nOut; ++o) { v[o] = (float)h->gsMean[o]; v[o + nOut] = (float)h->gsRms[o]; }
      fwrite(v.data(), sizeof(float), v.size(), f); fclose(f);
    }
    h->gsCalls++;
  }
  h->nStep++;   // AdamOptimizer::prepare_update (Optimizer.cpp:119)
  // Learner::processMemoryBuffer (Learner.cpp:74-100) up to the counters all-reduce; none of
  // it reads the (possibly still in flight) gradient sum, so doing it here keeps the order
  const int64_t currStep = h->nGradSteps + 1;
  updateTrainingStatistics(h);
  h->momentsPending = false;
  if (currStep % 1000 == 0) {
    computeMoments(h, h->moments);
    if (h->cfg.n_ranks > 1) h->momentsPending = true; else applyMoments(h, h->moments, false, 10);
  }
  applyEpisodesRemoval(h);
  h->inStep = true; return HL_OK;
}
int ol_moments_exchange(ol_learner* h, double* io, int32_t write_back) {
  if (!h || !io) return HL_ERR_BAD_ARG;
  if (!h->momentsPending) return fail(h, HL_ERR_STATE, "no reward/state moments pending this step");
  const size_t n = h->moments.size();
  if (write_back) for (size_t i = 0; i < n; ++i) h->moments[i] = io[i];
  else for (size_t i = 0; i < n; ++i) io[i] = (double)h->moments[i];
  return HL_OK;
}
int ol_grad_exchange(ol_learner* h, float* grad_io, int32_t write_back) {
  if (!h || !grad_io) return HL_ERR_BAD_ARG;
  if (write_back) std::copy(grad_io, grad_io + h->nParams, h->G.begin());
  else std::copy(h->G.begin(), h->G.end(), grad_io);
  return HL_OK;
}
int ol_counters_exchange(ol_learner* h, int64_t c[4], int32_t write_back) {
  if (!h || !c) return HL_ERR_BAD_ARG;
  if (write_back) { h->seenEpsGlobal = c[0]; h->seenStepsGlobal = c[1]; h->nFarGlobal = c[2]; h->nStoredGlobal = c[3]; h->countersReduced = true; }
  else { c[0] = h->nSeenEps; c[1] = h->nSeenSteps; c[2] = h->stats.nFarPolicySteps; c[3] = h->nTransitions; }
  return HL_OK;
}
int ol_step_end(ol_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  if (!h->inStep) return fail(h, HL_ERR_STATE, "step_end without step_begin");
  if (h->momentsPending) { applyMoments(h, h->moments, false, 10); h->momentsPending = false; }
  updateCounters(h, false);
  adamApply(h);                                   // Learner_approximator::applyGradient (:99-105)
  h->nGradSteps++;                                // Learner::globalGradCounterUpdate (Learner.cpp:130-133)
  h->inStep = false; return HL_OK;
}
int ol_step(ol_learner* h, int32_t n, const int64_t* flat) {
  if (!h) return HL_ERR_BAD_ARG;
  for (int s = 0; s < n; ++s) {
    int rc = ol_step_begin(h, flat ? flat + (size_t)s * h->B : nullptr); if (rc) return rc;
    rc = ol_step_end(h); if (rc) return rc;
  }
  return HL_OK;
}
// Episode::packEpisode / unpackEpisode (ReplayMemory/Episode.cpp:24-130, sizes Episode.h:211-228)
int64_t ol_packed_episode_size(const ol_learner* h, int32_t N) {
  if (!h || N < 0) return -1;
  return (int64_t)(h->dS + h->dA + h->polDim + 1 + 6) * N + 10;
}
int ol_append_packed_episode(ol_learner* h, const float* data, int64_t n) {
  if (!h || !data) return HL_ERR_BAD_ARG;
  const int dS = h->dS, dA = h->dA, pD = h->polDim, tup = dS + 1 + dA + pD;
  const int64_t N = (n - 10) / (tup + 6);
  if (N < 2 || ol_packed_episode_size(h, (int32_t)N) != n) return fail(h, HL_ERR_BAD_ARG, "packed episode has the wrong size");
  std::vector<float> S((size_t)N * dS), V(N), ADV(N);
  std::vector<double> A((size_t)N * dA), MU((size_t)N * pD), R(N);
  const float* buf = data;
  for (int64_t i = 0; i < N; ++i) {
    std::copy(buf, buf + dS, S.begin() + i * dS); R[i] = buf[dS]; buf += dS + 1;
    for (int j = 0; j < dA; ++j) A[i * dA + j] = buf[j]; buf += dA;
    for (int j = 0; j < pD; ++j) MU[i * pD + j] = buf[j]; buf += pD;
  }
  buf += N;                                            // returnEstimator: recomputed on insertion
  std::copy(buf, buf + N, ADV.begin()); buf += N;      // actionAdvantage
  std::copy(buf, buf + N, V.begin()); buf += N;        // stateValue
  buf += 3 * N;                                        // deltaValue, offPolicImpW, KullbLeibDiv: reset on insertion
  const char* cp = reinterpret_cast<const char*>(buf);
  bool term; int64_t ID; std::memcpy(&term, cp, sizeof(bool)); std::memcpy(&ID, cp + sizeof(bool), sizeof(int64_t));
  return ol_append_episode(h, (int32_t)N, S.data(), A.data(), MU.data(), R.data(), V.data(), ADV.data(), term ? 1 : 0, ID);
}
int ol_pack_episode(ol_learner* h, int64_t pos, float* dst, int64_t cap) {
  if (!h || !dst || pos < 0 || pos >= (int64_t)h->episodes.size()) return HL_ERR_BAD_ARG;
  const Episode& EP = *h->episodes[(size_t)pos];
  const int dS = h->dS, dA = h->dA; const int64_t N = EP.N;
  if (cap < ol_packed_episode_size(h, (int32_t)N)) return HL_ERR_BAD_ARG;
  std::fill(dst, dst + ol_packed_episode_size(h, (int32_t)N), 0.f);
  float* buf = dst;
  for (int64_t i = 0; i < N; ++i) {
    std::copy(EP.S.begin() + i * dS, EP.S.begin() + (i + 1) * dS, buf); buf[dS] = (float)EP.R[i]; buf += dS + 1;
    for (int j = 0; j < dA; ++j) buf[j] = (float)EP.A[i * dA + j]; buf += dA;
    for (int j = 0; j < h->polDim; ++j) buf[j] = (float)EP.MU[i * h->polDim + j]; buf += h->polDim;
  }
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.RET[i]; buf += N;
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.ADV[i]; buf += N;
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.V[i]; buf += N;
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.DQ[i]; buf += N;
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.IMPW[i]; buf += N;
  for (int64_t i = 0; i < N; ++i) buf[i] = EP.DKL[i]; buf += N;
  char* cp = reinterpret_cast<char*>(buf);
  const bool term = EP.term; const int64_t ID = EP.tag, sampled = -1, agentID = 0;
  std::memcpy(cp, &term, sizeof(bool)); cp += sizeof(bool);
  std::memcpy(cp, &ID, 8); cp += 8; std::memcpy(cp, &sampled, 8); cp += 8; std::memcpy(cp, &agentID, 8);
  return HL_OK;
}

// Network::save / restart (Network/Network.cpp:22-68) over Layer::save of each type
// (Layer_Base.h:143-166, Layers.h:401-420, 554-567): compact fp32, no SIMD padding
static size_t packedSize(const ol_learner* h) {
  size_t n = 0;
  for (const Layer& l : h->layers) {
    if (l.type == L_DENSE) n += (size_t)l.size * (l.nIn + (l.rec ? l.size : 0) + 1);   // Layer_Base.h:143-153
    else if (l.type == L_PARAMRES) n += 2 * (size_t)l.size;
    else if (l.type == L_PARAM) n += (size_t)l.size;
    else if (l.type == L_LSTM || l.type == L_MGU) n += (size_t)actSize(l) * (l.nIn + l.size + 1);   // Layer_LSTM.h:186-197, Layer_GRU.h:248-258
    else if (l.type == L_CONV) n += (size_t)(l.nW + l.nB);                                          // Layer_Conv2D.h:215-231
  }
  return n;
}
static void packBlob(const ol_learner* h, const std::vector<nnReal>& P, std::vector<float>& out) {
  out.clear();
  for (const Layer& l : h->layers) {
    const nnReal* W = P.data() + l.indW; const nnReal* Bv = P.data() + l.indB;
    if (l.type == L_DENSE) {
      for (int i = 0; i < l.nIn + (l.rec ? l.size : 0); ++i) for (int o = 0; o < l.size; ++o) out.push_back((float)W[o + (int64_t)l.nOutSimd * i]);
      for (int o = 0; o < l.size; ++o) out.push_back((float)Bv[o]);
    } else if (l.type == L_PARAMRES) {
      for (int o = 0; o < l.size; ++o) out.push_back((float)W[o]);
      for (int o = 0; o < l.size; ++o) out.push_back((float)Bv[o]);
    } else if (l.type == L_PARAM) for (int o = 0; o < l.size; ++o) out.push_back((float)Bv[o]);
    else if (l.type == L_LSTM || l.type == L_MGU) {   // weights, then biases, as they lie
      for (int64_t w = 0; w < (int64_t)actSize(l) * (l.nIn + l.size); ++w) out.push_back((float)W[w]);
      for (int o = 0; o < actSize(l); ++o) out.push_back((float)Bv[o]);
    } else if (l.type == L_CONV) {                    // weights, then biases, as they lie
      for (int64_t w = 0; w < l.nW; ++w) out.push_back((float)W[w]);
      for (int64_t o = 0; o < l.nB; ++o) out.push_back((float)Bv[o]);
    }
  }
}
static void unpackBlob(const ol_learner* h, const std::vector<float>& in, std::vector<nnReal>& P) {
  size_t k = 0;
  for (const Layer& l : h->layers) {
    nnReal* W = P.data() + l.indW; nnReal* Bv = P.data() + l.indB;
    if (l.type == L_DENSE) {
      for (int i = 0; i < l.nIn + (l.rec ? l.size : 0); ++i) for (int o = 0; o < l.size; ++o) W[o + (int64_t)l.nOutSimd * i] = (nnReal)in[k++];
      for (int o = 0; o < l.size; ++o) Bv[o] = (nnReal)in[k++];
    } else if (l.type == L_PARAMRES) {
      for (int o = 0; o < l.size; ++o) W[o] = (nnReal)in[k++];
      for (int o = 0; o < l.size; ++o) Bv[o] = (nnReal)in[k++];
    } else if (l.type == L_PARAM) for (int o = 0; o < l.size; ++o) Bv[o] = (nnReal)in[k++];
    else if (l.type == L_LSTM || l.type == L_MGU) {
      for (int64_t w = 0; w < (int64_t)actSize(l) * (l.nIn + l.size); ++w) W[w] = (nnReal)in[k++];
      for (int o = 0; o < actSize(l); ++o) Bv[o] = (nnReal)in[k++];
    } else if (l.type == L_CONV) {
      for (int64_t w = 0; w < l.nW; ++w) W[w] = (nnReal)in[k++];
      for (int64_t o = 0; o < l.nB; ++o) Bv[o] = (nnReal)in[k++];
    }
  }
}
int ol_save(ol_learner* h, const char* base) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  const std::vector<nnReal>* blobs[3] = {&h->W, &h->M1, &h->M2};
  const char* suf[3] = {"_weights", "_1stMom", "_2ndMom"};
  std::vector<float> buf;
  for (int b = 0; b < 3; ++b) {
    packBlob(h, *blobs[b], buf);
    const std::string name = std::string(base) + suf[b] + ".raw";
    FILE* f = fopen(name.c_str(), "wb");
    if (!f) return fail(h, HL_ERR_IO, "cannot write checkpoint file");
    fwrite(buf.data(), sizeof(float), buf.size(), f); fclose(f);
  }
  return HL_OK;
}
int ol_restart(ol_learner* h, const char* base) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  std::vector<nnReal>* blobs[3] = {&h->W, &h->M1, &h->M2};
  const char* suf[3] = {"_weights", "_1stMom", "_2ndMom"};
  const size_t n = packedSize(h);
  for (int b = 0; b < 3; ++b) {
    const std::string name = std::string(base) + suf[b] + ".raw";
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) { if (b == 0) return fail(h, HL_ERR_IO, "Parameters restart file not found"); continue; }
    std::vector<float> buf(n + 1);
    const size_t got = fread(buf.data(), sizeof(float), n + 1, f); fclose(f);
    if (got != n) return fail(h, HL_ERR_IO, "Mismatch in restarted file");
    buf.resize(n); unpackBlob(h, buf, *blobs[b]);
  }
  return HL_OK;
}
int ol_sync(ol_learner*) { return HL_OK; }
int ol_prepare_steps(ol_learner* h, int32_t n) { return (h && n >= 1) ? HL_OK : HL_ERR_BAD_ARG; }   // (a launch-shape hint of the device library)
// Approximator::forward(agent) (Network/Approximator.h:300-330) on standardised states
// (Episode::standardizedState, Episode.h:172-183): the network outputs RACER::selectAction reads
int ol_forward(ol_learner* h, int32_t n, const float* states, double* outputs) {
  if (!h || n < 0 || (n > 0 && (!states || !outputs))) return HL_ERR_BAD_ARG;
  if (h->cfg.nn_type != HL_NN_FFNN) return fail(h, HL_ERR_UNSUPPORTED, "forward of a recurrent net needs the agent's history");
  // with appended observations `states` holds, per row, the raw state of step t followed by those of t-1 .. t-nAppendedObs
  const int dS1 = h->dS, dS = dS1 * (1 + h->cfg.nAppendedObs), nOut = h->nOut;
  std::vector<nnReal> inp(dS);
  for (int r = 0; r < n; ++r) {
    for (int i = 0; i < dS; ++i) inp[i] = (states[(size_t)r * dS + i] - h->stMean[i % dS1]) * h->stScale[i % dS1];
    forwardNet(h, inp.data(), h->X, h->Y);
    getOutput(h, h->Y, outputs + (size_t)r * nOut);
  }
  return HL_OK;
}
// MemoryBuffer::agentToMinibatch (:440-467) + Approximator::forward(agent): the last steps from a zero recurrent state
int ol_forward_sequence(ol_learner* h, int32_t nSteps, const float* states, double* outputs) {
  if (!h || nSteps < 1 || !states || !outputs) return HL_ERR_BAD_ARG;
  const int dS = h->dS;
  if (h->cfg.nn_type == HL_NN_FFNN) return ol_forward(h, 1, states + (size_t)(nSteps - 1) * dS, outputs);
  // appended observations: up to nAppendedObs further states may be given in front of the window of min(nnBPTTseq, t) + 1 steps; they
  // only feed the first steps' appended slots, and steps before the first given one repeat it (Episode::standardizedState's intent)
  const int nApp = h->cfg.nAppendedObs, recK = (h->cfg.nnBPTTseq > 0 ? h->cfg.nnBPTTseq : 16) + 1;
  const int win = std::min((int)nSteps, recK), ctx = nSteps - win;
  if (ctx > nApp) return fail(h, HL_ERR_BAD_ARG, "more steps than nnBPTTseq + 1 (+ nAppendedObs)");
  std::vector<Act> series((size_t)win);
  std::vector<nnReal> inp((size_t)dS * (1 + nApp));
  for (int k = 0; k < win; ++k) {
    series[k].X = h->X; series[k].Y = h->Y;
    for (int j = 0; j <= nApp; ++j) { const int g = std::max(ctx + k - j, 0);
      for (int i = 0; i < dS; ++i) inp[(size_t)j * dS + i] = (states[(size_t)g * dS + i] - h->stMean[i]) * h->stScale[i]; }
    forwardNet(h, inp.data(), series[k].X, series[k].Y, k ? &series[k - 1].Y : nullptr);
  }
  getOutput(h, series[win - 1].Y, outputs);
  return HL_OK;
}
int ol_grad_stats(ol_learner* h, double* mean, double* rms) {
  if (!h || !mean || !rms) return HL_ERR_BAD_ARG;
  if (h->gsMean.empty()) return fail(h, HL_ERR_STATE, "no gradient step yet");
  std::copy(h->gsMean.begin(), h->gsMean.end(), mean); std::copy(h->gsRms.begin(), h->gsRms.end(), rms);
  return HL_OK;
}
int ol_set_episode_log(ol_learner* h, const char* path) { if (!h) return HL_ERR_BAD_ARG; h->episodeLog = path ? path : ""; return HL_OK; }
int ol_set_log_base(ol_learner* h, const char* base) { if (!h) return HL_ERR_BAD_ARG; h->logBase = base ? base : ""; return HL_OK; }
int ol_set_tap(ol_learner* h, int32_t e) { if (!h) return HL_ERR_BAD_ARG; h->tap = e != 0; return HL_OK; }

int ol_readback(ol_learner* h, int32_t what, void* dst, int64_t bytes) {
  if (!h || !dst) return HL_ERR_BAD_ARG;
  const void* src = nullptr; int64_t n = 0;
  switch (what) {
    case HL_TAP_FLAT: src = h->bFlat.data(); n = h->bFlat.size() * 8; break;
    case HL_TAP_EPISODE: src = h->bEp.data(); n = h->bEp.size() * 8; break;
    case HL_TAP_TSTEP: src = h->bT.data(); n = h->bT.size() * 8; break;
    case HL_TAP_TAG: src = h->bTag.data(); n = h->bTag.size() * 8; break;
    case HL_TAP_STATE: src = h->tState.data(); n = h->tState.size() * 4; break;
    case HL_TAP_OUTPUT: src = h->tO.data(); n = h->tO.size() * 8; break;
    case HL_TAP_OUTGRAD: src = h->tG.data(); n = h->tG.size() * 8; break;
    case HL_TAP_RHO: src = h->tRho.data(); n = h->tRho.size() * 8; break;
    case HL_TAP_DKL: src = h->tDkl.data(); n = h->tDkl.size() * 8; break;
    case HL_TAP_DELTAQ: src = h->tDq.data(); n = h->tDq.size() * 8; break;
    case HL_TAP_FAR: src = h->tFar.data(); n = h->tFar.size(); break;
    case HL_TAP_GRADSUM: src = h->tGradSum.data(); n = h->tGradSum.size() * 4; break;
    default: return HL_ERR_BAD_ARG;
  }
  if (n == 0) return fail(h, HL_ERR_STATE, "tap not recorded (hl_set_tap before the step)");
  if (bytes < n) return HL_ERR_BAD_ARG;
  std::memcpy(dst, src, n); return HL_OK;
}
int ol_get_scalars(ol_learner* h, hl_scalars* o) {
  if (!h || !o) return HL_ERR_BAD_ARG;
  o->beta = h->beta; o->alpha = h->alpha; o->CmaxRet = h->CmaxRet; o->CinvRet = h->CinvRet;
  o->nGradSteps = h->nGradSteps; o->nStoredSteps = h->nTransitions; o->nStoredEps = h->episodes.size();
  o->nFarPolicySteps = h->stats.nFarPolicySteps; o->nSeenSteps = h->seenStepsUpd; o->nSeenEps = h->seenEpsUpd;
  o->adam_beta_t_1 = h->beta_t_1; o->adam_beta_t_2 = h->beta_t_2; o->adam_nStep = h->nStep;
  return HL_OK;
}
// MemoryProcessing::histogramImportanceWeights (MemoryProcessing.cpp:353-389) with Utilities::real2SS (SstreamUtilities.h:51-63)
static void realToSS(std::ostringstream& B, const double V, const int W, const bool bPos) {
  B << " " << std::setw(W);
  if (std::fabs(V) >= 1e4) B << std::setprecision(std::max(W - 7 + bPos, 0));
  else if (std::fabs(V) >= 1e3) B << std::setprecision(std::max(W - 6 + bPos, 0));
  else if (std::fabs(V) >= 1e2) B << std::setprecision(std::max(W - 5 + bPos, 0));
  else if (std::fabs(V) >= 1e1) B << std::setprecision(std::max(W - 4 + bPos, 0));
  else B << std::setprecision(std::max(W - 3 + bPos, 0));
  B << std::fixed << V;
}
// MemoryBuffer::getMetrics / getHeaders (ReplayMemory/MemoryBuffer.cpp:522-575) followed by AdamOptimizer::getMetrics /
// getHeaders (Network/Optimizer.cpp:216-226), in the order Learner::processStats calls them (Learners/Learner.cpp:158-196):
// the metrics first -- they consume the return-estimate counters --, then the header
int ol_metrics(ol_learner* h, char* header, int32_t headerCap, char* line, int32_t lineCap) {
  if (!h) return HL_ERR_BAD_ARG;
  hl_stats& st = h->stats;
  const bool qStats = st.minQ < st.maxQ;
  if (line) {
    std::ostringstream buff;
    realToSS(buff, st.avgReturn, 9, 0); realToSS(buff, (double)h->rewMean, 6, 0); realToSS(buff, (double)h->rewStd, 6, 1);
    realToSS(buff, st.avgKLdivergence, 5, 1);
    if (qStats) {
      const Real EPS = std::numeric_limits<float>::epsilon();
      st.avgSquaredErr = std::max(EPS, st.avgSquaredErr);
      realToSS(buff, std::sqrt(st.avgSquaredErr), 6, 1); realToSS(buff, st.maxAbsError, 6, 1);
      if (st.countReturnsEstimateUpdates > 0) {
        const int64_t nRet = std::max((int64_t)1, st.countReturnsEstimateUpdates);
        const Real eRet = std::max(EPS, st.sumReturnsEstimateErrors);
        realToSS(buff, std::sqrt(eRet / nRet), 6, 1);
        st.countReturnsEstimateUpdates = 0; st.sumReturnsEstimateErrors = 0;
      } else { st.countReturnsEstimateUpdates = -1; st.sumReturnsEstimateErrors = 0; }
      realToSS(buff, st.stdevQ, 6, 1); realToSS(buff, st.avgQ, 6, 0); realToSS(buff, st.minQ, 6, 0); realToSS(buff, st.maxQ, 6, 0);
    }
    buff << " " << std::setw(5) << (long)h->episodes.size();
    buff << " " << std::setw(7) << (long)h->nTransitions;
    buff << " " << std::setw(7) << (long)h->seenEpsUpd;
    buff << " " << std::setw(8) << (long)h->seenStepsUpd;
    buff << " " << std::setw(7) << (long)st.nFarPolicySteps;
    if (h->CmaxRet > 1) realToSS(buff, h->beta, 6, 1);
    long double sum = 0; for (nnReal x : h->W) sum += (long double)x * (long double)x;
    realToSS(buff, (double)std::sqrt(sum), 7, 1);
    const std::string sLine = buff.str();
    if ((int)sLine.size() + 1 > lineCap) return HL_ERR_BAD_ARG;
    std::memcpy(line, sLine.c_str(), sLine.size() + 1);
  }
  if (header) {
    std::ostringstream buff;
    buff << "|  avgR  | avgr | stdr | DKL ";
    if (qStats) {
      if (st.countReturnsEstimateUpdates >= 0) buff << "| RMSE |maxErr| dRet | stdQ | avgQ | minQ | maxQ ";
      else buff << "| RMSE |maxErr| stdQ | avgQ | minQ | maxQ ";
    }
    buff << "| nEp |  nObs | totEp | totObs | nFarP ";
    if (h->CmaxRet > 1) buff << "| beta ";
    buff << std::left << std::setfill(' ') << "| " << std::setw(6) << "net";
    const std::string sHead = buff.str();
    if ((int)sHead.size() + 1 > headerCap) return HL_ERR_BAD_ARG;
    std::memcpy(header, sHead.c_str(), sHead.size() + 1);
  }
  return HL_OK;
}
int ol_impweight_histogram(ol_learner* h, char* text, int32_t cap, int64_t counts[HL_IMPW_BINS]) {
  if (!h) return HL_ERR_BAD_ARG;
  constexpr int nBins = 81;
  const Real beg = std::log(1e-3), end = std::log(50.0);
  Fval bounds[nBins + 1] = {0}; int64_t cnt[nBins] = {0};
  for (int i = 1; i < nBins; ++i) bounds[i] = std::exp(beg + (end - beg) * (i - 1.0) / (nBins - 2.0));
  bounds[nBins] = std::numeric_limits<Fval>::max() - 1e2;
  for (const auto& ep : h->episodes) for (int j = 0; j < ep->ndata(); ++j) {
    const Fval rho = ep->IMPW[j];
    for (int b = 0; b < nBins; ++b) if (rho >= bounds[b] && rho < bounds[b + 1]) cnt[b]++;
  }
  if (counts) std::memcpy(counts, cnt, sizeof(cnt));
  if (text) {
    std::ostringstream buff;
    buff << "_____________________________________________________________________";
    buff << "\nOFF-POLICY IMP WEIGHTS HISTOGRAMS\n";
    buff << "weight pi/mu (harmonic mean of histogram's bounds):\n";
    for (int b = 0; b < nBins; ++b) { const Fval x = bounds[b], y = bounds[b + 1]; realToSS(buff, 2 * x * (y / (x + y)), 6, 1); }
    buff << "\nfraction of dataset:\n";
    const Real dataSize = (Real)h->nTransitions;
    for (int b = 0; b < nBins; ++b) realToSS(buff, cnt[b] / dataSize, 6, 1);
    buff << "\n";
    buff << "_____________________________________________________________________";
    const std::string t = buff.str();
    if ((int)t.size() + 1 > cap) return fail(h, HL_ERR_BAD_ARG, "text buffer too small");
    std::memcpy(text, t.c_str(), t.size() + 1);
  }
  return HL_OK;
}
int ol_get_counts(ol_learner* h, int64_t* nStoredSteps, int64_t* nStoredEps, int64_t* nGradSteps, int64_t* nSeenSteps, int64_t* nSeenEps) {
  if (!h) return HL_ERR_BAD_ARG;
  if (nStoredSteps) *nStoredSteps = h->nTransitions; if (nStoredEps) *nStoredEps = (int64_t)h->episodes.size();
  if (nGradSteps) *nGradSteps = h->nGradSteps; if (nSeenSteps) *nSeenSteps = h->nSeenSteps; if (nSeenEps) *nSeenEps = h->nSeenEps;
  return HL_OK;
}
int ol_get_initial_data(ol_learner* h, int64_t* n) { if (!h || !n) return HL_ERR_BAD_ARG; *n = h->nGatheredB4Startup; return HL_OK; }
int ol_get_stats(ol_learner* h, hl_stats* o) { if (!h || !o) return HL_ERR_BAD_ARG; *o = h->stats; return HL_OK; }

int ol_synth_episode_len(const synth_cfg* c, uint64_t e, int* term) { return synth_episode_len(c, e, term); }
void ol_synth_episode(const synth_cfg* c, uint64_t e, float* s, double* a, double* mu, double* r, float* v) {
  synth_episode(c, e, s, a, mu, r, v);
}
void ol_synth_episode_discrete(const synth_cfg* c, int nOpt, uint64_t e, float* s, double* a, double* mu, double* r, float* v) {
  synth_episode_discrete(c, nOpt, e, s, a, mu, r, v);
}

}  // extern "C"
