/*
 * oracle/ref_driver.cpp -- harness around the UNMODIFIED reference learner.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference source: it
 * #includes the reference headers where they lie under /root/reference and is
 * linked against objects compiled from the reference .cpp files in place
 * (oracle/Makefile; outputs only under oracle/_ref/).  It
 *   (1) instantiates RACER<Zero_advantage,Continuous_policy,Rvec> (= VRACER,
 *       Learners/AlgoFactory.cpp:132-152) exactly as Appendix B of SURVEY.md,
 *   (2) fills its MemoryBuffer with the deterministic episodes of
 *       oracle/synth.h (through Episode's public fields + finalize +
 *       computeReturnEstimator + pushBackEpisode, i.e. the calls
 *       MemoryBuffer::addEpisodeToTrainingSet makes, MemoryBuffer.cpp:131-170)
 *       or through the plug-in path Learner::select (mode bench fill=select),
 *   (3) either steps the learner through its own task queue ("official") or
 *       re-plays the body of Learner_approximator::spawnTrainTasks
 *       (Learner_approximator.cpp:36-92) + RACER::setupTasks stepMain /
 *       stepComplete (RACER.cpp:81-108) call by call ("manual") so that
 *       per-sample quantities can be tapped between the reference's own calls,
 *   (4) writes golden fixtures (oracle/blobio.h format) or timing JSON.
 * "official" and "manual" produce bit-identical weights (tests check this).
 */
#define protected public
#define private public
#include "smarties/Learners/RACER.h"
#include "smarties/Math/Zero_advantage.h"
#include "smarties/Math/Gaus_advantage.h"
#include "smarties/Math/Discrete_policy.h"
#include "smarties/Math/Discrete_advantage.h"
#include "smarties/Math/Continuous_policy.h"
#include "smarties/Network/Approximator.h"
#include "smarties/Network/Optimizer.h"
#include "smarties/Network/Network.h"
#include "smarties/ReplayMemory/MemoryProcessing.h"
#include "smarties/Utils/TaskQueue.h"
#undef protected
#undef private

#include "synth.h"
#include "blobio.h"

#include <chrono>
#include <unistd.h>
#include <cstdio>
#include <map>
#include <set>
#include <sstream>
#include <string>

using namespace smarties;

// Runs with several learners (mpiexec -n 2 ref_driver fixture out.bin ...; learners_train_comm = the world): the reference polls its
// delayed reductions with MPI_Test right behind their MPI_Iallreduce (DelayedReductor::get(false), Utils/DelayedReductor.cpp:36-48;
// MemoryProcessing::updateCounters :56-58, updateRewardsStats :147-150), so a step uses this step's global sums or the previous
// step's, whichever the network's timing gives.  The harness pins the timing through MPI's profiling interface, without touching the
// reference: with gPromptReductions every poll finds its reduction complete (the order of operations of SURVEY.md 8(e)).
// prompt=2 pins the OTHER end of that freedom (round 6): no poll of a delayed reduction ever finds it complete -- it completes in the
// MPI_Wait of the next update() -- so every step uses the PREVIOUS step's global sums (the first one the start-up sums), the 1000th
// step's statistics update the start-up moments once more.  Only the delayed reductors' requests are held back (they reduce MPI_LONG /
// MPI_LONG_DOUBLE, tagged where they are issued); the gradient's MPI_Test (Optimizer.h:110-116) is left alone or the step would never end.
static int gPromptReductions = 0;
static std::set<MPI_Request> gDelayedRequests;
extern "C" int MPI_Iallreduce(const void* sb, void* rb, int count, MPI_Datatype dt, MPI_Op op, MPI_Comm comm, MPI_Request* req) {
  const int rc = PMPI_Iallreduce(sb, rb, count, dt, op, comm, req);
  if (gPromptReductions == 2 && (dt == MPI_LONG || dt == MPI_LONG_DOUBLE)) gDelayedRequests.insert(*req);
  return rc;
}
extern "C" int MPI_Wait(MPI_Request* req, MPI_Status* st) {
  if (gPromptReductions == 2) gDelayedRequests.erase(*req);
  return PMPI_Wait(req, st);
}
extern "C" int MPI_Test(MPI_Request* req, int* flag, MPI_Status* st) {
  if (gPromptReductions == 2) {
    if (gDelayedRequests.count(*req)) { *flag = 0; return MPI_SUCCESS; }
    return PMPI_Test(req, flag, st);
  }
  if (gPromptReductions) { *flag = 1; return PMPI_Wait(req, st); }
  return PMPI_Test(req, flag, st);
}
static int worldSize() { int n = 1; MPI_Comm_size(MPI_COMM_WORLD, &n); return n; }
static int worldRank() { int r = 0; MPI_Comm_rank(MPI_COMM_WORLD, &r); return r; }

struct Args {
  std::map<std::string, std::string> kv;
  std::string s(const std::string& k, const std::string& d) const {
    auto it = kv.find(k); return it == kv.end() ? d : it->second;
  }
  double d(const std::string& k, double dflt) const {
    auto it = kv.find(k); return it == kv.end() ? dflt : atof(it->second.c_str());
  }
  long l(const std::string& k, long dflt) const {
    auto it = kv.find(k); return it == kv.end() ? dflt : atol(it->second.c_str());
  }
};

static std::vector<Uint> parseList(const std::string& s) {
  std::vector<Uint> r; std::stringstream ss(s); std::string tok;
  while (std::getline(ss, tok, ',')) if (tok.size()) r.push_back((Uint)atol(tok.c_str()));
  return r;
}

// conv=W,H,C,K,F,S;...: the arguments of Communicator::setPreprocessingConv2d (Communicator.cpp:136-162), one group per layer
static std::vector<Conv2D_Descriptor> parseConv(const std::string& s) {
  std::vector<Conv2D_Descriptor> r; std::stringstream ss(s); std::string grp;
  while (std::getline(ss, grp, ';')) {
    const std::vector<Uint> v = parseList(grp);
    if (v.size() != 6) continue;
    Conv2D_Descriptor d;
    d.inpX = v[0]; d.inpY = v[1]; d.inpFeatures = v[2]; d.outFeatures = v[3];
    d.filterx = d.filtery = v[4]; d.stridex = d.stridey = v[5]; d.paddinx = d.paddiny = 0;
    d.outY = (d.inpY - d.filterx + 2 * d.paddinx) / d.stridex + 1;
    d.outX = (d.inpX - d.filtery + 2 * d.paddiny) / d.stridey + 1;
    r.push_back(d);
  }
  return r;
}

// Sampler of the harness for MDPs with appended observations: Episode::standardizedState (Episode.h:172-183) reads
// states[samp - j] through an unsigned subtraction, i.e. out of bounds for samp < nAppendedObs, so the fixtures draw
// (episode, step >= minT) pairs only -- from a generator of the harness, the learner's own is left to the Adam draws.
// Output sorted by (episode, step) and unique, like Sample_uniform's (Sampling.cpp:82-96).
struct RestrictedSampler : public Sampling {
  std::mt19937 g; Uint minT;
  RestrictedSampler(std::vector<std::mt19937>& G, MemoryBuffer* R, Uint minT_, unsigned seed) : Sampling(G, R, false), g(seed), minT(minT_) {}
  void sample(std::vector<Uint>& seq, std::vector<Uint>& obs) override {
    std::set<std::pair<Uint, Uint>> S;
    const Uint nE = episodes.size();
    while (S.size() < seq.size()) {
      const Uint e = g() % nE, nd = episodes[e]->ndata();
      if (nd <= minT) continue;
      S.insert({e, minT + (Uint)(g() % (nd - minT))});
    }
    Uint i = 0; for (const auto& p : S) { seq[i] = p.first; obs[i] = p.second; ++i; }
  }
  void prepare() override {}
  bool requireImportanceWeights() override { return false; }
};

// large parameter vectors in "lean" fixtures: every 53rd element, plus the sum and the sum of squares in double
static bool gLean = false;
static void writeParams(BlobWriter& W, const std::string& name, const std::vector<float>& v) {
  if (!gLean || name == "s1_gradSum") { W.f32(name, v); return; }      // (the first gradient stays whole)
  std::vector<float> sub; double s1 = 0, s2 = 0;
  for (size_t i = 0; i < v.size(); ++i) { if (i % 53 == 0) sub.push_back(v[i]); s1 += v[i]; s2 += (double)v[i] * v[i]; }
  W.f32(name + "_sub", sub); W.f64(name + "_sums", std::vector<double>{s1, s2, (double)v.size()});
}

static std::vector<uint32_t> rngState(const std::mt19937& g) {
  std::ostringstream ss; ss << g;
  std::istringstream is(ss.str());
  std::vector<uint32_t> r; unsigned long v;
  while (is >> v) r.push_back((uint32_t)v);
  return r;  // 624 state words + position
}

// -DREF_GAUSS_ADV builds the same harness around RACER with the Gaussian advantage head
// (RACER<Param_advantage = Gaussian_advantage, Continuous_policy, Rvec>, Learners/AlgoFactory.cpp:124)
#if defined(REF_DISCRETE)
// -DREF_DISCRETE: RACER<Discrete_advantage, Discrete_policy, Uint> (Learners/AlgoFactory.cpp:109), one action
// variable with `nOpt` options
using VRACER = RACER<Discrete_advantage, Discrete_policy, Uint>;
using POLICY = Discrete_policy;
static const char* kLearner = "RACER";
static const int64_t kAdvKind = 2;
#elif defined(REF_GAUSS_ADV)
using VRACER = RACER<Gaussian_advantage, Continuous_policy, Rvec>;
using POLICY = Continuous_policy;
static const char* kLearner = "RACER";
static const int64_t kAdvKind = 1;
#else
using VRACER = RACER<Zero_advantage, Continuous_policy, Rvec>;
using POLICY = Continuous_policy;
static const char* kLearner = "VRACER";
static const int64_t kAdvKind = 0;
#endif

struct Harness {
  ExecutionInfo& info;
  MDPdescriptor MDP;
  std::unique_ptr<HyperParameters> HP;
  std::unique_ptr<VRACER> L;
  std::unique_ptr<TaskQueue> algo, dataQ;
  synth_cfg SC;
  int nOpt = 0;     // discrete harness: number of action options

  Harness(ExecutionInfo& I, const Args& A) : info(I) {
    const int nThr = (int)A.l("threads", 1);
    info.nThreads = nThr; omp_set_num_threads(nThr);
    info.randSeed = A.l("seed", 42); info.initialze();
    info.learners_train_comm = MPI_COMM_SELF; info.bIsMaster = true;
    if (worldSize() > 1) {      // every rank a learner with its own share of the batch and of the replay (HyperParameters::defineDistributedLearning)
      MPI_Comm c; MPI_Comm_dup(MPI_COMM_WORLD, &c); info.learners_train_comm = c;
      gPromptReductions = (int)A.l("prompt", 1);
    }
    info.nAgents = 1; info.nOwnedEnvironments = 1; info.nEnvironments = 1;
    info.logAllSamples = (int)A.l("rewlog", 0); info.learnersOnWorkers = false; info.restart = "none";
    if (info.logAllSamples) {     // cumulative_rewards.dat / obs.raw of MemoryBuffer::pushBackEpisode go to a scratch directory
      snprintf(info.initial_runDir, sizeof(info.initial_runDir), "%s", A.s("rewdir", "/tmp").c_str());
    }
    const Uint dS = A.l("dimS", 17), dA = A.l("dimA", 6);
    MDP.dimState = dS; MDP.dimAction = dA;
    const std::string bnd = A.s("bounded", std::string(dA, '1'));
    MDP.bActionSpaceBounded = std::vector<bool>(dA, false);
    for (Uint i = 0; i < dA && i < bnd.size(); ++i) MDP.bActionSpaceBounded[i] = bnd[i] == '1';
#ifdef REF_DISCRETE
    nOpt = (int)A.l("nOpt", 4);
    if (dA != 1) { fprintf(stderr, "discrete harness: dimA must be 1\n"); exit(2); }
    MDP.discreteActionValues = std::vector<Uint>(1, (Uint)nOpt);
#endif
    MDP.nAppendedObs = (Uint)A.l("nApp", 0);
    MDP.isPartiallyObservable = A.l("pomdp", 0) != 0;      // with nnType left at FFNN: RNN encoder layers under MGU layers (Approximator.cpp:221-223, 264-270)
    MDP.conv2dDescriptors = parseConv(A.s("conv", ""));
    MDP.synchronize([](void*, size_t) {});
#ifdef REF_DISCRETE
    MDP.policyVecDim = nOpt;
#else
    MDP.policyVecDim = 2 * dA;
#endif
    HP = std::make_unique<HyperParameters>(dS, dA);
    HP->learner = kLearner; HP->returnsEstimator = A.s("retEst", "retrace");      // retrace | retraceExplore | GAE | none (MemoryProcessing.cpp:418-450)
    HP->nnOutputFunc = A.s("nnOutputFunc", "Linear");                             // activation of the output layer (Approximator.cpp:193,228)
    HP->encoderLayerSizes = parseList(A.s("encoder", ""));                        // Learner_approximator::createEncoder (:149-166)
    HP->nnLayerSizes = parseList(A.s("layers", "256,256"));
    HP->nnFunc = A.s("nnFunc", "SoftSign");
    HP->nnType = A.s("nnType", "FFNN"); HP->nnBPTTseq = (Uint)A.l("bptt", 16);   // "LSTM": recurrent hidden layers
    HP->batchSize = A.l("batch", 256);
    HP->maxTotObsNum = A.l("maxObs", 1000000);
    HP->minTotObsNum = A.l("minObs", HP->maxTotObsNum);
    HP->clipImpWeight = A.d("clip", 4);
    HP->penalTol = A.d("penalTol", 0.1);
    HP->epsAnneal = A.d("epsAnneal", 0);
    HP->gamma = A.d("gamma", 0.995);
    HP->lambda = A.d("lambda", 1);
    HP->learnrate = A.d("learnrate", 1e-4);
    HP->explNoise = A.d("explNoise", 0.4472135955);
    HP->outWeightsPrefac = A.d("outWeightsPrefac", 0.1);
    HP->nnLambda = A.d("nnLambda", 0);
    HP->dataSamplingAlgo = A.s("sampling", "uniform");   // uniform | PERrank | PERerr | PERseq (Sampling.cpp:298-340)
    HP->ERoldSeqFilter = A.s("erFilter", "oldest");      // oldest | farpolfrac | maxkldiv | minerror (MemoryProcessing.cpp:261-298)
    HP->obsPerStep = 0;  // never block gradient steps on data
    HP->saveFreq = 1000000000;
    HP->defineDistributedLearning(info); HP->check();
    L = std::make_unique<VRACER>(MDP, *HP, info);
    L->setLearnerName("agent_00", 0);
    algo = std::make_unique<TaskQueue>([]() { return false; });
    dataQ = std::make_unique<TaskQueue>([]() { return false; });
    L->setupTasks(*algo); L->setupDataCollectionTasks(*dataQ);
    if (MDP.nAppendedObs > 0 || A.kv.count("minT")) const_cast<std::unique_ptr<Sampling>&>(L->data->sampler) = std::make_unique<RestrictedSampler>(info.generators, L->data.get(), (Uint)A.l("minT", MDP.nAppendedObs), (unsigned)A.l("sampleSeed", 99));      // (minT = nApp + bptt for recurrent nets: the window's first steps are read the same way)
    SC.seed = (uint64_t)A.l("synthSeed", 7); SC.dimS = (int)dS; SC.dimA = (int)dA;
    SC.lenMin = (int)A.l("lenMin", 201); SC.lenMax = (int)A.l("lenMax", 201);
    SC.pTerminated = A.d("pTerm", 0.0); SC.muSpread = A.d("muSpread", 0.5);
    SC.actNoise = A.d("actNoise", 1.0);
  }

  // mirror of MemoryBuffer::addEpisodeToTrainingSet for a ready-made episode
  void pushSynthEpisode(uint64_t e) {
    int term = 0; const int N = synth_episode_len(&SC, e, &term);
    const int dS = SC.dimS, dA = SC.dimA;
    std::vector<float> S((size_t)N * dS), V(N);
#ifdef REF_DISCRETE
    const int pD = nOpt;
    std::vector<double> Act((size_t)N * dA), Mu((size_t)N * pD), R(N);
    synth_episode_discrete(&SC, nOpt, e, S.data(), Act.data(), Mu.data(), R.data(), V.data());
#else
    const int pD = 2 * dA;
    std::vector<double> Act((size_t)N * dA), Mu((size_t)N * pD), R(N);
    synth_episode(&SC, e, S.data(), Act.data(), Mu.data(), R.data(), V.data());
#endif
    auto EP = std::make_unique<Episode>(MDP);
    EP->bReachedTermState = term; EP->agentID = (Sint)e;  // agentID doubles as content tag
    for (int t = 0; t < N; ++t) {
      EP->states.push_back(Fvec(S.begin() + (size_t)t * dS, S.begin() + (size_t)(t + 1) * dS));
      EP->latent_states.push_back(Fvec());
      EP->actions.push_back(Rvec(Act.begin() + (size_t)t * dA, Act.begin() + (size_t)(t + 1) * dA));
      EP->policies.push_back(Rvec(Mu.begin() + (size_t)t * pD, Mu.begin() + (size_t)(t + 1) * pD));
      EP->rewards.push_back(R[t]);
      if (t) EP->totR += R[t];
      EP->stateValue.push_back(V[t]);
      EP->actionAdvantage.push_back(0);
      if (t && t < N - 1) L->data->increaseLocalSeenSteps();   // storeAction (MemoryBuffer.cpp:110)
    }
    const long tStamp = std::max(L->data->nLocTimeStepsTrain(), (long)0);
    EP->finalize(tStamp);
    MemoryProcessing::computeReturnEstimator(*L->data, *EP);
    L->data->pushBackEpisode(std::move(EP));
    L->data->increaseLocalSeenSteps();                         // MemoryBuffer.cpp:167
    L->data->increaseLocalSeenEps();
  }
};

static double now_s() {
  return std::chrono::duration<double>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

struct StepTap {
  std::vector<int64_t> flat, tag, tstep;
  std::vector<double> O, G, rho, dkl, dq;   // double-precision taps
  std::vector<uint8_t> far;
};

// One gradient step, call by call as the reference makes them, with taps.
static void manualStep(Harness& H, StepTap* tap) {
  VRACER& L = *H.L;
  Approximator& NET = *L.networks[0];
  const Uint B = H.HP->batchSize_local;
  L.profiler->stop();
  const MiniBatch MB = L.data->sampleMinibatch(B, L.nGradSteps());
  const Uint nOut = NET.nOutputs();
  if (tap) {
    // episode order at sampling time -> flat index of each sample
    std::map<const Episode*, int64_t> prefix; int64_t p = 0;
    for (Uint k = 0; k < L.data->episodes.size(); ++k) {
      prefix[L.data->episodes[k].get()] = p; p += L.data->episodes[k]->ndata();
    }
    tap->flat.resize(B); tap->tag.resize(B); tap->tstep.resize(B);
    tap->O.resize(B * nOut); tap->G.resize(B * nOut);
    tap->rho.resize(B); tap->dkl.resize(B); tap->dq.resize(B); tap->far.resize(B);
    for (Uint b = 0; b < B; ++b) {
      tap->tstep[b] = MB.sampledTstep(b);
      tap->tag[b] = MB.episodes[b]->agentID;
      tap->flat[b] = prefix[MB.episodes[b]] + MB.sampledTstep(b);
    }
  }
  for (Uint b = 0; b < B; ++b) {
    NET.load(MB, b, 0);
    const Uint t = MB.sampledTstep(b);
    const Real beta = L.beta, Cmax = L.CmaxRet, Cinv = L.CinvRet;
    L.Train(MB, 0, b);
    if (tap) {
      const Activation* A = NET.getContext(b).activation(t, 0);
      const Rvec O = A->getOutput();
      const NNvec G = A->getOutputDelta();
      for (Uint o = 0; o < nOut; ++o) { tap->O[b * nOut + o] = O[o]; tap->G[b * nOut + o] = G[o]; }
      const POLICY POL(L.pol_start, L.aInfo, O);
      const Real RHO = POL.importanceWeight(MB.action(b, t), MB.mu(b, t));
      tap->rho[b] = RHO; tap->dkl[b] = POL.KLDivergence(MB.mu(b, t));
      tap->dq[b] = MB.episodes[b]->deltaValue[t];
      tap->far[b] = isFarPolicy(RHO, Cmax, Cinv) ? 1 : 0;
      (void)beta;
    }
    NET.backProp(b);
  }
  NET.prepareUpdate();
  NET.updateGradStats(L.learner_name, L.nGradSteps());
}

static void finishStep(Harness& H) {
  VRACER& L = *H.L;
  L.processMemoryBuffer();
  L.logStats();
  L.applyGradient();
  L.globalGradCounterUpdate();
}

static std::vector<float> paramsOf(const Parameters* P) {
  return std::vector<float>(P->params, P->params + P->nParams);
}

static int modeFixture(ExecutionInfo& info, const Args& A, const std::string& out) {
  Harness H(info, A);
  VRACER& L = *H.L;
  gLean = A.l("lean", 0) != 0;
  const long nEps = A.l("nEps", 40), nSteps = A.l("nSteps", 10), tapSteps = A.l("tapSteps", nSteps), addEvery = A.l("addEvery", 0);
  const std::vector<Uint> gradSteps = parseList(A.s("gradSteps", "1,2"));
  const std::vector<Uint> retSteps = parseList(A.s("retSteps", ""));
  const bool official = A.s("path", "manual") == "official";
  const int nRanks = worldSize(), rank = worldRank();
  for (long e = 0; e < nEps; ++e) if (e % nRanks == rank) H.pushSynthEpisode((uint64_t)e);      // (round robin over the learners)
  if (rank == 0) std::remove((L.learner_name + "_stats.txt").c_str());      // (the run's own <learner>_stats.txt is captured below)
  // resume=<prefix>: instead of (or on top of) a synthetic fill, what Learner_approximator::restart does (Learner_approximator.cpp:
  // 118-131) with the files an earlier run of this harness wrote through ckpt=<prefix> memck=<prefix>: network and Adam moments,
  // replay memory with its counters and scaling, the optimizer's step count
  const std::string resume = A.s("resume", "");
  if (!resume.empty()) {
    for (const auto& net : L.networks) net->restart(resume);
    L.data->restart(resume);
    for (const auto& net : L.networks) net->setNgradSteps(L.nGradSteps());
  }

  BlobWriter W(nRanks > 1 ? out + ".r" + std::to_string(rank) : out);
  if (nRanks > 1) W.i64("ranks", std::vector<int64_t>{nRanks, rank, gPromptReductions});
  Approximator& NET = *L.networks[0];
  AdamOptimizer* OPT = dynamic_cast<AdamOptimizer*>(NET.opt.get());
  const Parameters* PW = NET.net->weights.get();
  {
    std::vector<int64_t> cfg = {(int64_t)H.MDP.dimStateObserved, (int64_t)H.MDP.dimAction,
        (int64_t)H.HP->batchSize, nEps, nSteps, (int64_t)PW->nParams, (int64_t)NET.nOutputs(),
        (int64_t)L.data->nStoredSteps(), (int64_t)H.SC.seed, H.SC.lenMin, H.SC.lenMax, kAdvKind, (int64_t)H.nOpt,
        (int64_t)(H.HP->nnType == "LSTM" ? 1 : (H.HP->nnType == "MGU" ? 2 : (H.HP->nnType == "RNN" ? 3 : 0))), (int64_t)H.HP->nnBPTTseq};
    W.i64("cfg", cfg);
    {   // MDP preprocessing: appended observations and the convolutional layers (W, H, C, K, F, S per layer)
      std::vector<int64_t> pre = {(int64_t)H.MDP.nAppendedObs};
      for (const auto& d : H.MDP.conv2dDescriptors) for (int64_t v : {(int64_t)d.inpX, (int64_t)d.inpY, (int64_t)d.inpFeatures, (int64_t)d.outFeatures, (int64_t)d.filterx, (int64_t)d.stridex}) pre.push_back(v);
      W.i64("preproc", pre);
      const std::string f = H.HP->ERoldSeqFilter;
      W.i64("erFilter", std::vector<int64_t>{f == "farpolfrac" ? 1 : (f == "maxkldiv" ? 2 : (f == "minerror" ? 3 : 0))});
      const std::string sa = H.HP->dataSamplingAlgo;
      W.i64("sampling", std::vector<int64_t>{sa == "PERrank" ? 1 : (sa == "PERerr" ? 2 : (sa == "PERseq" ? 3 : 0))});
      W.i64("threads", std::vector<int64_t>{(int64_t)H.info.nThreads});
      W.i64("addEvery", std::vector<int64_t>{(int64_t)A.l("addEvery", 0)});
      W.i64("minObs", std::vector<int64_t>{(int64_t)H.HP->minTotObsNum});
      const std::string re = H.HP->returnsEstimator;
      W.i64("retEst", std::vector<int64_t>{re == "retraceExplore" ? 1 : (re == "GAE" ? 2 : (re == "none" ? 3 : 0))});
      const char* fn[] = {"Linear", "Tanh", "SoftSign", "Relu", "LRelu", "Sigm", "HardSign", "SoftPlus", "ExpPlus", "Exp"};
      int64_t of = 0; for (int i = 0; i < 10; ++i) if (H.HP->nnOutputFunc == fn[i]) of = i;
      W.i64("outFunc", std::vector<int64_t>{of});
      std::vector<int64_t> enc; for (auto v : H.HP->encoderLayerSizes) enc.push_back((int64_t)v);
      W.i64("encoder", enc);
      if (H.MDP.isPartiallyObservable) W.i64("pomdp", std::vector<int64_t>{1});
    }
    std::vector<int64_t> lay; for (auto v : H.HP->nnLayerSizes) lay.push_back((int64_t)v);
    W.i64("layers", lay);
    std::vector<uint8_t> bnd; for (Uint i = 0; i < H.MDP.dimAction; ++i) bnd.push_back(H.MDP.bActionSpaceBounded[i]);
    W.u8("bounded", bnd);
    std::vector<double> hp = {H.HP->clipImpWeight, H.HP->penalTol, H.HP->epsAnneal, H.HP->gamma,
        H.HP->lambda, H.HP->learnrate, H.HP->explNoise, H.HP->outWeightsPrefac, H.HP->nnLambda,
        H.SC.pTerminated, H.SC.muSpread, H.SC.actNoise, (double)H.HP->maxTotObsNum};
    W.f64("hp", hp);
    std::vector<int64_t> iw, ib, nw, nb;
    for (Uint l = 0; l < PW->nLayers; ++l) { iw.push_back(PW->indWeights[l]); ib.push_back(PW->indBiases[l]);
      nw.push_back(PW->nWeights[l]); nb.push_back(PW->nBiases[l]); }
    W.i64("indWeights", iw); W.i64("indBiases", ib); W.i64("nWeights", nw); W.i64("nBiases", nb);
  }
  writeParams(W, "W0", paramsOf(PW));
  W.u32("rng_before_init", rngState(info.generators[0]));

  // stepInit (RACER.cpp:69-79): Learner::initializeLearner
  L.initializeLearner(); L.algoSubStepID = 0; L.profiler->start("DATA");
  W.u32("rng0", rngState(info.generators[0]));
  if (A.l("pack", 0)) {   // Episode::packEpisode (ReplayMemory/Episode.cpp:24-86) of the first stored episodes: the
    // worker -> learner wire format and the per-episode record of MemoryBuffer::save
    const long nPack = std::min<long>(A.l("pack", 0), (long)L.data->nStoredEps());
    std::vector<int64_t> ids;
    for (long k = 0; k < nPack; ++k) {
      Episode& EP = L.data->get(k);
      ids.push_back((int64_t)EP.agentID);                       // the harness keeps the synthetic episode number here
      W.f32("pack_" + std::to_string(k), EP.packEpisode());
    }
    W.i64("pack_tags", ids);
  }
  W.scalar_d("beta0", L.data->beta); W.scalar_d("cmax0", L.data->CmaxRet);
  {
    std::vector<float> sc;
    for (auto v : H.MDP.stateMean) sc.push_back(v);
    for (auto v : H.MDP.stateScale) sc.push_back(v);
    sc.push_back(H.MDP.rewardsMean); sc.push_back(H.MDP.rewardsScale); sc.push_back(H.MDP.rewardsStdDev);
    W.f32("scaling0", sc);
    // Retrace estimates after the initial rescale, keyed by content tag
    std::vector<int64_t> tags, off; std::vector<float> ret; int64_t o = 0;
    for (Uint k = 0; k < L.data->episodes.size(); ++k) {
      const Episode& EP = *L.data->episodes[k];
      tags.push_back(EP.agentID); off.push_back(o); o += EP.nsteps();
      ret.insert(ret.end(), EP.returnEstimator.begin(), EP.returnEstimator.end());
    }
    W.i64("ret0_tags", tags); W.i64("ret0_off", off); W.f32("ret0", ret);
  }

  std::vector<double> traj_beta, traj_cmax, traj_wnorm; std::vector<int64_t> traj_nfar;
  for (long k = 1; k <= nSteps; ++k) {
    const std::string sk = "s" + std::to_string(k) + "_";
    if (official) {
      while (L.nGradSteps() < k) H.algo->run();
    } else {
      const bool bTap = k <= tapSteps;
      if (bTap) {
        W.u32(sk + "rng", rngState(info.generators[0]));
        std::vector<int64_t> order;
        for (Uint i = 0; i < L.data->episodes.size(); ++i) order.push_back(L.data->episodes[i]->agentID);
        W.i64(sk + "order", order);
        W.scalar_d(sk + "beta", L.data->beta); W.scalar_d(sk + "cmax", L.data->CmaxRet);
      }
      StepTap tap;
      manualStep(H, bTap ? &tap : nullptr);
      if (bTap) {
        const int64_t B = tap.flat.size(), nO = NET.nOutputs();
        W.i64(sk + "flat", tap.flat); W.i64(sk + "tag", tap.tag); W.i64(sk + "t", tap.tstep);
        W.f64(sk + "O", tap.O, {B, nO}); W.f64(sk + "G", tap.G, {B, nO});
        W.f64(sk + "rho", tap.rho); W.f64(sk + "dkl", tap.dkl); W.f64(sk + "dq", tap.dq);
        W.u8(sk + "far", tap.far);
      }
      for (auto g : gradSteps) if ((long)g == k) writeParams(W, sk + "gradSum", paramsOf(OPT->gradSum.get()));
      finishStep(H);
    }
    // episodes that arrive while the learner trains (addEvery = n: one more synthetic episode behind every n-th step, as the
    // data-collection tasks of a run would deliver them): time stamps, placeholder errors and removals of a replay in motion
    if (addEvery > 0 && k % addEvery == 0) H.pushSynthEpisode((uint64_t)(nEps + k / addEvery - 1));
    for (auto g : gradSteps) if ((long)g == k) {
      writeParams(W, sk + "W", paramsOf(PW));
      writeParams(W, sk + "M1", paramsOf(OPT->_1stMom.get())); writeParams(W, sk + "M2", paramsOf(OPT->_2ndMom.get()));
    }
    for (auto g : retSteps) if ((long)g == k) {
      std::vector<int64_t> tags; std::vector<float> ret, val, impw, dkl, dq;
      for (Uint i = 0; i < L.data->episodes.size(); ++i) {
        const Episode& EP = *L.data->episodes[i];
        tags.push_back(EP.agentID);
        ret.insert(ret.end(), EP.returnEstimator.begin(), EP.returnEstimator.end());
        val.insert(val.end(), EP.stateValue.begin(), EP.stateValue.end());
        impw.insert(impw.end(), EP.offPolicImpW.begin(), EP.offPolicImpW.end());
        dkl.insert(dkl.end(), EP.KullbLeibDiv.begin(), EP.KullbLeibDiv.end());
        dq.insert(dq.end(), EP.deltaValue.begin(), EP.deltaValue.end());
      }
      W.i64(sk + "ep_tags", tags); W.f32(sk + "ret", ret); W.f32(sk + "val", val);
      W.f32(sk + "impw", impw); W.f32(sk + "ep_dkl", dkl); W.f32(sk + "ep_dq", dq);
      std::vector<float> sc;
      for (auto v : H.MDP.stateMean) sc.push_back(v);
      for (auto v : H.MDP.stateScale) sc.push_back(v);
      sc.push_back(H.MDP.rewardsMean); sc.push_back(H.MDP.rewardsScale); sc.push_back(H.MDP.rewardsStdDev);
      W.f32(sk + "scaling", sc);
    }
    traj_beta.push_back(L.data->beta); traj_cmax.push_back(L.data->CmaxRet);
    traj_nfar.push_back((int64_t)L.data->nFarPolicySteps());
    traj_wnorm.push_back((double)PW->compute_weight_norm());
  }
  W.f64("traj_beta", traj_beta); W.f64("traj_cmax", traj_cmax);
  W.i64("traj_nfar", traj_nfar); W.f64("traj_wnorm", traj_wnorm);
  writeParams(W, "Wfinal", paramsOf(PW));
  writeParams(W, "M1final", paramsOf(OPT->_1stMom.get())); writeParams(W, "M2final", paramsOf(OPT->_2ndMom.get()));
  {   // the reference's own checkpoint of this state (Approximator::save -> AdamOptimizer::save -> Network::save)
    const std::string ck = A.s("ckpt", "");
    if (!ck.empty()) {
      NET.save(ck, false);
      for (const char* suf : {"_net_weights", "_net_1stMom", "_net_2ndMom"}) {
        FILE* f = fopen((ck + suf + ".raw").c_str(), "rb");
        std::vector<uint8_t> bytes;
        if (f) { int c; while ((c = fgetc(f)) != EOF) bytes.push_back((uint8_t)c); fclose(f); }
        W.u8(std::string("ckpt") + suf, bytes);
      }
    }
  }
  {   // the reference's own replay-memory checkpoint of this state (MemoryBuffer::save, MemoryBuffer.cpp:274-324)
    const std::string mk = A.s("memck", "");
    if (!mk.empty()) {
      L.data->save(mk);
      for (const char* suf : {"_scaling", "_rank_000_learner_status", "_rank_000_learner_data"}) {
        FILE* f = fopen((mk + suf + ".raw").c_str(), "rb");
        std::vector<uint8_t> bytes;
        if (f) { int c; while ((c = fgetc(f)) != EOF) bytes.push_back((uint8_t)c); fclose(f); }
        W.u8(std::string("memck") + suf, bytes);
      }
    }
  }
  {   // the statistics line and header of agent_00_stats.txt for this state (Learner::logStats, Learner.cpp:155-195)
    const ReplayStats& st0 = L.data->stats;
    std::vector<double> s0 = {(double)st0.avgKLdivergence, (double)st0.avgSquaredErr, (double)st0.maxAbsError,
        (double)st0.avgReturn, (double)st0.avgQ, (double)st0.stdevQ, (double)st0.minQ, (double)st0.maxQ, (double)st0.nFarPolicySteps};
    W.f64("stats_before_metrics", s0);
    std::ostringstream buf, head;
    L.data->getMetrics(buf); L.getMetrics(buf);
    L.data->getHeaders(head); L.getHeaders(head);
    const std::string sb = buf.str(), sh = head.str();
    W.u8("metrics_line", std::vector<uint8_t>(sb.begin(), sb.end()));
    W.u8("metrics_head", std::vector<uint8_t>(sh.begin(), sh.end()));
  }
  {   // <learner>_stats.txt as the run wrote it (Learner::processStats, Learner.cpp:158-196: header + one line per 1000 steps)
    FILE* f = fopen((L.learner_name + "_stats.txt").c_str(), "rb");
    std::vector<uint8_t> bytes;
    if (f) { int c; while ((c = fgetc(f)) != EOF) bytes.push_back((uint8_t)c); fclose(f); }
    W.u8("stats_file", bytes);
  }
  {   // output-gradient statistics (Utils/StatsTracker.cpp): the file the run wrote at iter % 1000 == 0 and the
      // mean / RMS over the last minibatch
    FILE* f = fopen((L.learner_name + "_net_outGrad_stats.raw").c_str(), "rb");
    std::vector<uint8_t> bytes;
    if (f) { int c; while ((c = fgetc(f)) != EOF) bytes.push_back((uint8_t)c); fclose(f); }
    W.u8("outgrad_stats_file", bytes);
    const StatsTracker* T = NET.gradStats;
    std::vector<double> inst;
    for (Uint i = 0; i < T->n_stats; ++i) inst.push_back((double)T->instMean[i]);
    for (Uint i = 0; i < T->n_stats; ++i) inst.push_back((double)T->instStdv[i]);
    W.f64("outgrad_stats_last", inst);
  }
  if (A.l("rewlog", 0)) {   // what the reference wrote for every episode that entered the training set (MemoryBuffer.cpp:491-513)
    const std::string dir = A.s("rewdir", "/tmp");
    const std::string fr = dir + "/agent_00_rank_000_cumulative_rewards.dat", fo = dir + "/agent_00_rank_000_obs.raw";
    std::vector<uint8_t> bytes; FILE* g = fopen(fr.c_str(), "rb");
    if (g) { int c; while ((c = fgetc(g)) != EOF) bytes.push_back((uint8_t)c); fclose(g); }
    remove(fr.c_str()); remove(fo.c_str());
    W.u8("rewards_log", bytes);
  }
  if (A.l("hist", 0)) {   // the importance-weight histogram Learner::logStats prints with the profiler (Learner.cpp:139-144):
    // MemoryProcessing::histogramImportanceWeights writes to stdout only, which is pointed at a file for the call
    fflush(stdout);
    const std::string tmp = out + ".hist.txt";
    const int saved = dup(fileno(stdout));
    FILE* f = fopen(tmp.c_str(), "w"); dup2(fileno(f), fileno(stdout));
    MemoryProcessing::histogramImportanceWeights(*L.data);
    fflush(stdout); dup2(saved, fileno(stdout)); close(saved); fclose(f);
    std::vector<uint8_t> bytes; FILE* g = fopen(tmp.c_str(), "rb");
    if (g) { int c; while ((c = fgetc(g)) != EOF) bytes.push_back((uint8_t)c); fclose(g); }
    remove(tmp.c_str());
    W.u8("impw_histogram", bytes);
  }
  {
    const ReplayStats& st = L.data->stats;
    std::vector<double> s = {(double)st.avgKLdivergence, (double)st.avgSquaredErr, (double)st.maxAbsError,
        (double)st.avgReturn, (double)st.avgQ, (double)st.stdevQ, (double)st.minQ, (double)st.maxQ,
        (double)st.nFarPolicySteps};
    W.f64("stats_final", s);
  }
  printf("fixture %s: nParams %lu nObs %ld steps %ld beta %.9g wnorm %.9Lg\n", out.c_str(),
         (unsigned long)PW->nParams, L.data->nStoredSteps(), nSteps, L.data->beta, PW->compute_weight_norm());
  return 0;
}

static int modeBench(ExecutionInfo& info, const Args& A) {
  Harness H(info, A);
  VRACER& L = *H.L;
  const long nObs = A.l("nObs", 1000000), nSteps = A.l("nSteps", 100), warm = A.l("warmup", 10);
  const double t0 = now_s();
  if (A.s("fill", "synth") == "select") {
    Agent AG(0, 0, 0, H.MDP); AG.initializeActionSampling(info.generators[0]);
    std::mt19937 g(7); std::normal_distribution<double> N(0, 1);
    while (L.locDataSetSize() < nObs) {
      std::vector<double> s(H.MDP.dimState); for (auto& x : s) x = N(g);
      AG.update(INIT, s, 0.0); L.select(AG);
      for (int t = 1; t <= 200; ++t) { for (auto& x : s) x = N(g);
        AG.update(t == 200 ? LAST : CONT, s, N(g)); L.select(AG); }
    }
  } else {
    uint64_t e = 0;
    while (L.locDataSetSize() < nObs) H.pushSynthEpisode(e++);
  }
  const double t1 = now_s();
  H.algo->run();  // stepInit (+ first step)
  while (L.nGradSteps() < warm) H.algo->run();
  const long g0 = L.nGradSteps(); const double t2 = now_s();
  while (L.nGradSteps() < g0 + nSteps) H.algo->run();
  const double dt = now_s() - t2;
  printf("{\"kind\":\"reference\",\"threads\":%d,\"nObs\":%ld,\"nEps\":%ld,\"steps\":%ld,\"seconds\":%.6f,"
         "\"steps_per_s\":%.3f,\"transitions_per_s\":%.1f,\"fill_s\":%.3f,\"batch\":%lu,\"beta\":%.9g}\n",
         (int)info.nThreads, L.locDataSetSize(), L.data->nStoredEps(), nSteps, dt, nSteps / dt,
         nSteps * (double)H.HP->batchSize / dt, t1 - t0, (unsigned long)H.HP->batchSize, L.data->beta);
  if (A.l("profile", 0)) printf("%s\n", L.profiler->printStatAndReset().c_str());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: ref_driver fixture <out.bin> k=v... | bench k=v...\n"); return 2; }
  const std::string mode = argv[1];
  Args A; std::string out;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i]; const auto p = a.find('=');
    if (p == std::string::npos) out = a; else A.kv[a.substr(0, p)] = a.substr(p + 1);
  }
  int one = 1; char* av[] = {argv[0], nullptr}; char** avp = av;
  ExecutionInfo info(one, avp);
  if (mode == "fixture") return modeFixture(info, A, out);
  if (mode == "bench") return modeBench(info, A);
  fprintf(stderr, "unknown mode\n"); return 2;
}
