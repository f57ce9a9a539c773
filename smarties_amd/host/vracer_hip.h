// smarties_amd/host/vracer_hip.h -- C++ host side above the C-ABI (include/smarties_hip.h).
//
// A header-only mirror of the part of the reference's Learner interface that drives the hot path,
// with the reference's names, argument meaning and error behaviour, so that a maintainer can lift
// it into `Learners/RACER_HIP.{h,cpp}` and so that the C++ parity test
// (tests/cpp/host_parity.cpp) reads like a smarties test:
//
//   Learner::select                 Learners/Learner.cpp:30-45
//   RACER::selectAction             Learners/RACER.cpp:30-47
//   RACER::processTerminal          Learners/RACER.cpp:49-59
//   Learner::initializeLearner      Learners/Learner.cpp:47-72
//   RACER::setupTasks (train step)  Learners/RACER.cpp:62-110
//   Approximator::save / restart    Network/Approximator.cpp:282-297
//   Continuous_policy::selectAction Math/Continuous_policy.h:777-789 (per-dimension Normal /
//   SquashedNormal sample with sampleClippedGaussian, :183-197, 347-363)
//
// The reference `die()`s on errors (Utils/Warnings.h:34-44: message + abort); here `die` throws
// std::runtime_error with the library's message so that a test can observe it.
//
// Not reproduced here: the smarties::Communicator / Worker machinery that produces `Agent`s, and
// action rescaling to the environment's bounds (ActionInfo::action2scaledAction) -- both stay the
// reference's; `Agent` below carries only what Learner::select reads and writes.
#pragma once
#include <cmath>
#include <cstdint>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include <limits>
#include "../../include/smarties_hip.h"

namespace smarties_amd {

using Real = double;
using Fval = float;
using Uint = std::size_t;
using Rvec = std::vector<Real>;
using Fvec = std::vector<Fval>;

enum episodeStatus { INIT = 0, CONT, LAST, TERM, FAIL };   // Core/Agent.h:23

// Core/StateAction.h:57-112 (the fields the learner reads)
struct MDPdescriptor {
  Uint dimStateObserved = 0, dimAction = 0;
  std::vector<bool> bActionSpaceBounded;
  std::vector<Uint> discreteActionValues;     // non-empty: discrete actions (one variable is served), options per variable
  bool bDiscreteActions() const { return discreteActionValues.size() > 0; }
};

// Settings/HyperParameters.h:42-72 (same names, same defaults where they do not depend on the MDP)
struct HyperParameters {
  Real explNoise = std::sqrt(0.2), gamma = 0.995, lambda = 1, clipImpWeight = 4, penalTol = 0.1;
  Real epsAnneal = 5e-7, nnLambda = 0, learnrate = 1e-4, outWeightsPrefac = 1e-3;
  Uint minTotObsNum = 0, maxTotObsNum = 1 << 20, batchSize = 256;
  std::vector<Uint> nnLayerSizes = {128, 128};
  std::string nnFunc = "Tanh";
  std::string nnType = "FFNN";      // "FFNN", "LSTM", "MGU" / "GRU" or "RNN" (Network/Builder.cpp:48-117); nnBPTTseq: steps of truncated BPTT
  Uint nnBPTTseq = 16;
  std::string nnOutputFunc = "Linear";          // activation of the output layer (Network/Approximator.cpp:193,228)
  std::vector<Uint> encoderLayerSizes = {};     // dense layers of the preprocessing network, ahead of nnLayerSizes (Learner_approximator.cpp:149-166)
  std::string returnsEstimator = "default";     // "default" / "retrace", "retraceExplore", "GAE", "none" (MemoryProcessing.cpp:418-450)
  std::string learner = "VRACER";   // "VRACER" (Zero_advantage) or "RACER" (Gaussian_advantage), AlgoFactory.cpp:109-152
  std::string ERoldSeqFilter = "oldest";        // "oldest", "farpolfrac", "maxkldiv", "minerror" (MemoryProcessing.cpp:261-298)
  std::string dataSamplingAlgo = "uniform";     // "uniform", "PERrank", "PERerr", "PERseq" (Sampling.cpp:298-340)
  Uint randSeed = 0;
};

// Core/Agent.h (subset): what Learner::select reads (state, reward, status) and writes (action, policy)
struct Agent {
  Uint ID = 0;
  episodeStatus agentStatus = INIT;
  Fvec state;                 // observed state, raw
  Real reward = 0;
  Rvec action, policyVector;  // policy-space action a_t and behaviour policy mu_t = [mean, stdev]
  std::mt19937 generator;
  explicit Agent(Uint id = 0, Uint seed = 0) : ID(id), generator((unsigned)(seed + 1000 * id)) {}
};

[[noreturn]] inline void die(const std::string& msg) { throw std::runtime_error(msg); }

class VRACER {
  hl_learner* H = nullptr;
  MDPdescriptor MDP;
  HyperParameters S;
  bool bTrain = true, bInit = false;
  bool recurrent = false;
  int nOut = 0, nDense = 0, nAdv = 0, nOpt = 0;      // nAdv: advantage outputs between V and the policy; nOpt: discrete options
  // one in-progress episode per agent: MemoryBuffer::inProgress (ReplayMemory/MemoryBuffer.h)
  struct InProgress { Fvec states; Rvec actions, policies, rewards; Fvec values, advantages; int64_t tag = 0; };
  std::vector<InProgress> inProgress;
  int64_t nSeenEpisodes = 0;

  void ck(int rc) const { if (rc) die(std::string(hl_status_string(rc)) + ": " + hl_last_error(H)); }
  static int funcId(const std::string& f) {
    if (f == "Tanh") return HL_FUNC_TANH;
    if (f == "SoftSign") return HL_FUNC_SOFTSIGN;
    if (f == "Relu") return HL_FUNC_RELU;
    if (f == "Linear") return HL_FUNC_LINEAR;
    if (f == "LRelu") return HL_FUNC_LRELU;
    if (f == "Sigm") return HL_FUNC_SIGM;
    if (f == "HardSign") return HL_FUNC_HARDSIGN;
    if (f == "SoftPlus") return HL_FUNC_SOFTPLUS;
    if (f == "ExpPlus") return HL_FUNC_EXPPLUS;
    if (f == "Exp") return HL_FUNC_EXP;
    die("Activation function not recognized");      // makeFunction (Network/Layers/Functions.h:643-668)
  }
  InProgress& episodeOf(const Agent& a) { if (a.ID >= inProgress.size()) inProgress.resize(a.ID + 1); return inProgress[a.ID]; }

 public:
  // Learners/RACER_common.cpp:23-27
  static Real scaleNet2V(const Real x) {
    return x > 0 ? 100 * (x + 51) - 100 * std::sqrt(2601 + 100 * x) : 100 * (x - 51) + 100 * std::sqrt(2601 - 100 * x);
  }
  // Math/Continuous_policy.h:183-190
  static Real sampleClippedGaussian(std::mt19937& gen) {
    constexpr Real NORMDIST_MAX = 3;   // Settings/Bund.h:51
    std::normal_distribution<Real> dist(0, 1);
    std::uniform_real_distribution<Real> safety(-NORMDIST_MAX, NORMDIST_MAX);
    const Real noise = dist(gen);
    if (noise > NORMDIST_MAX || noise < -NORMDIST_MAX) return safety(gen);
    return noise;
  }

  VRACER(const MDPdescriptor& M, const HyperParameters& hp, int deviceID = 0, int nLearners = 1, int learnerRank = 0)
      : MDP(M), S(hp) {
    if (M.dimAction > HL_MAX_DIMA || hp.nnLayerSizes.size() + hp.encoderLayerSizes.size() > HL_MAX_HIDDEN) die("problem too large for hl_config");
    hl_config c{}; c.struct_size = sizeof(c);
    c.dimS = (int32_t)M.dimStateObserved; c.dimA = (int32_t)M.dimAction;
    for (Uint i = 0; i < M.dimAction; ++i) c.bounded[i] = i < M.bActionSpaceBounded.size() && M.bActionSpaceBounded[i];
    c.n_hidden = (int32_t)hp.nnLayerSizes.size();
    for (Uint i = 0; i < hp.nnLayerSizes.size(); ++i) c.hidden[i] = (int32_t)hp.nnLayerSizes[i];
    c.nnFunc = funcId(hp.nnFunc);
    if (hp.nnType == "LSTM") { c.nn_type = HL_NN_LSTM; c.nnBPTTseq = (int32_t)hp.nnBPTTseq; recurrent = true; }
    else if (hp.nnType == "MGU" || hp.nnType == "GRU") { c.nn_type = HL_NN_MGU; c.nnBPTTseq = (int32_t)hp.nnBPTTseq; recurrent = true; }   // Builder.cpp:68-73
    else if (hp.nnType == "RNN") { c.nn_type = HL_NN_RNN; c.nnBPTTseq = (int32_t)hp.nnBPTTseq; recurrent = true; }       // Builder.cpp:76-81
    else if (hp.nnType != "FFNN") die("nnType " + hp.nnType + " is not served by the HIP library");      // ("Recurrent": recurrent layers without a BPTT window, HyperParameters.cpp:209)
    c.nnOutputFunc = funcId(hp.nnOutputFunc);
    c.n_encoder = (int32_t)hp.encoderLayerSizes.size();
    for (Uint i = 0; i < hp.encoderLayerSizes.size(); ++i) c.encoder[i] = (int32_t)hp.encoderLayerSizes[i];
    // AlgoFactory.cpp:134-135: "default" means Retrace for RACER / VRACER
    c.returnsEstimator = (hp.returnsEstimator == "default" || hp.returnsEstimator == "retrace") ? HL_RET_RETRACE :
                         hp.returnsEstimator == "retraceExplore" ? HL_RET_RETRACE_EXPLORE : hp.returnsEstimator == "GAE" ? HL_RET_GAE :
                         hp.returnsEstimator == "none" ? HL_RET_NONE : -1;
    if (c.returnsEstimator < 0) die("returnsEstimator " + hp.returnsEstimator + " not recognized");
    // AlgoFactory.cpp:78-152: discrete action spaces always get RACER<Discrete_advantage, Discrete_policy, Uint>
    if (M.bDiscreteActions()) {
      if (M.dimAction != 1 || M.discreteActionValues.size() != 1) die("one discrete action variable is served");
      c.adv_kind = HL_ADV_DISCRETE; c.n_options = (int32_t)M.discreteActionValues[0]; nOpt = c.n_options;
    } else if (hp.learner == "VRACER") c.adv_kind = HL_ADV_ZERO;
    else if (hp.learner == "RACER") c.adv_kind = HL_ADV_GAUSSIAN;
    else die("learner " + hp.learner + " is not served by the HIP library");
    nAdv = c.adv_kind == HL_ADV_GAUSSIAN ? 1 + 2 * (int)M.dimAction : nOpt;
    c.ERoldSeqFilter = hp.ERoldSeqFilter == "oldest" ? HL_ER_OLDEST : hp.ERoldSeqFilter == "farpolfrac" ? HL_ER_FARPOLFRAC :
                       hp.ERoldSeqFilter == "maxkldiv" ? HL_ER_MAXKLDIV : hp.ERoldSeqFilter == "minerror" ? HL_ER_MINERROR : -1;
    if (c.ERoldSeqFilter < 0) die("ERoldSeqFilter setting not recognized.");
    c.dataSamplingAlgo = hp.dataSamplingAlgo == "uniform" ? HL_SAMPLE_UNIFORM : hp.dataSamplingAlgo == "PERrank" ? HL_SAMPLE_PERRANK :
                         hp.dataSamplingAlgo == "PERerr" ? HL_SAMPLE_PERERR : hp.dataSamplingAlgo == "PERseq" ? HL_SAMPLE_PERSEQ : -1;
    if (c.dataSamplingAlgo < 0) die("Setting dataSamplingAlgo not recognized.");
    c.batchSize = (int32_t)hp.batchSize; c.maxTotObsNum = (int64_t)hp.maxTotObsNum; c.minTotObsNum = (int64_t)hp.minTotObsNum;
    c.gamma = hp.gamma; c.lambda = hp.lambda; c.clipImpWeight = hp.clipImpWeight; c.penalTol = hp.penalTol;
    c.epsAnneal = hp.epsAnneal; c.learnrate = hp.learnrate; c.nnLambda = hp.nnLambda; c.explNoise = hp.explNoise;
    c.outWeightsPrefac = hp.outWeightsPrefac; c.randSeed = hp.randSeed;
    c.n_ranks = nLearners; c.rank = learnerRank; c.device_id = deviceID; c.ref_threads = 1;
    const int rc = hl_create(&c, &H);
    if (rc) {       // (hl_create may hand back a handle that only carries the message)
      const std::string msg = std::string("hl_create: ") + hl_status_string(rc) + ": " + hl_last_error(H);
      if (H) { hl_destroy(H); H = nullptr; }
      die(msg);
    }
    if (hl_init_weights(H) != HL_OK) { const std::string msg = hl_last_error(H); hl_destroy(H); H = nullptr; die(msg); }
    nOut = hl_num_outputs(H); nDense = nOpt ? 1 + 2 * nOpt : 1 + nAdv + (int)M.dimAction;
  }
  ~VRACER() { if (H) hl_destroy(H); }
  VRACER(const VRACER&) = delete;
  VRACER& operator=(const VRACER&) = delete;

  hl_learner* handle() const { return H; }
  void setTrain(bool b) { bTrain = b; }

  // Approximator::forward(agent): network outputs for the agent's current state
  Rvec forward(const Agent& agent) const {
    if (agent.state.size() != MDP.dimStateObserved) die("Agent state has the wrong size");
    Rvec out((Uint)nOut);
    if (recurrent) {   // MemoryBuffer::agentToMinibatch (:440-467): the last min(nnBPTTseq, t) + 1 states of the episode in progress
      const InProgress& EP = inProgress.at(agent.ID);
      const Uint dS = MDP.dimStateObserved, nS = (Uint)(EP.states.size() / dS), n = std::min<Uint>(S.nnBPTTseq + 1, nS);
      if (n == 0) die("forward(agent) before the agent's state was stored");
      const int rcs = hl_forward_sequence(H, (int32_t)n, EP.states.data() + (size_t)(nS - n) * dS, out.data());
      if (rcs) die(std::string(hl_status_string(rcs)) + ": " + hl_last_error(H));
      return out;
    }
    const int rc = hl_forward(H, 1, agent.state.data(), out.data());
    if (rc) die(std::string(hl_status_string(rc)) + ": " + hl_last_error(H));
    return out;
  }

  // Learner::select (Learner.cpp:30-45): storeState; then act, or close the episode
  void select(Agent& agent) {
    InProgress& EP = episodeOf(agent);
    const Uint dS = MDP.dimStateObserved, dA = MDP.dimAction;
    if (agent.agentStatus == INIT) { EP = InProgress(); EP.tag = nSeenEpisodes++; }
    // MemoryBuffer::storeState (MemoryBuffer.cpp:79-129): the reward of the first state is 0
    EP.states.insert(EP.states.end(), agent.state.begin(), agent.state.end());
    EP.rewards.push_back(agent.agentStatus == INIT ? 0.0 : agent.reward);
    if (agent.agentStatus < LAST) {
      // RACER::selectAction (RACER.cpp:30-47)
      const Rvec output = forward(agent);
      if (nOpt) {     // Discrete_policy (Math/Discrete_policy.h:63-83, 190-200) + Discrete_advantage (:64-70)
        Rvec probs((Uint)nOpt); Real norm = 0;
        for (int j = 0; j < nOpt; ++j) { const Real x = output[(Uint)(1 + nOpt + j)]; probs[j] = (x + std::sqrt(1 + x * x)) / 2; norm += probs[j]; }
        norm = std::max(norm, std::numeric_limits<Real>::epsilon());
        for (int j = 0; j < nOpt; ++j) probs[j] /= norm;
        Uint label;
        if (bTrain) { std::discrete_distribution<Uint> dist(probs.begin(), probs.end()); label = dist(agent.generator); }
        else label = (Uint)(std::max_element(probs.begin(), probs.end()) - probs.begin());      // Utilities::maxInd
        Real expA = 0; for (int j = 0; j < nOpt; ++j) expA += probs[j] * output[(Uint)(1 + j)];
        const Real V = scaleNet2V(output[0]), A = output[1 + label] - expA;
        EP.values.push_back((Fval)V); EP.advantages.push_back((Fval)((Fval)(V + A) - (Fval)V));
        agent.action = Rvec(1, (Real)label + 0.1);                                              // label2actionMessage (StateAction.h:327-341)
        agent.policyVector = probs;
        EP.actions.push_back(agent.action[0]);
        EP.policies.insert(EP.policies.end(), probs.begin(), probs.end());
        return;
      }
      Rvec mean(dA), stdev(dA), act(dA);
      for (Uint i = 0; i < dA; ++i) {
        const Real p = output[(Uint)nDense + i];
        mean[i] = output[(Uint)(1 + nAdv) + i]; stdev[i] = (p + std::sqrt(1 + p * p)) / 2;         // SoftPlus (Functions.h:541-584)
        if (!bTrain) { act[i] = mean[i]; continue; }                                 // Continuous_policy.h:779
        const Real a = mean[i] + stdev[i] * sampleClippedGaussian(agent.generator);
        constexpr Real MAX = 8.31776613503286;                                       // SquashedNormalPolicy::sample (:356-360)
        act[i] = c_bounded(i) ? (a > MAX ? MAX : (a < -MAX ? -MAX : a)) : a;
      }
      const Real V = scaleNet2V(output[0]);
      // MB.appendValues(V, V + adv.computeAdvantage(action)) (RACER.cpp:44-46): Zero_advantage, or
      // Gaussian_advantage::computeAdvantage (Gaus_advantage.h:76-88) with the policy's clipped mean and variance
      Real A = 0;
      if (nAdv) {
        auto sp = [](Real x) { return (x + std::sqrt(1 + x * x)) / 2; };
        constexpr Real MAX = 8.31776613503286;
        Real quad = 0, ratio = 1;
        for (Uint i = 0; i < dA; ++i) {
          const Real m = c_bounded(i) ? (mean[i] > MAX ? MAX : (mean[i] < -MAX ? -MAX : mean[i])) : mean[i];
          const Real p1 = sp(output[2 + i]), p2 = sp(output[2 + dA + i]), Sv = stdev[i] * stdev[i];
          quad += (act[i] - m) * (act[i] - m) / (act[i] > m ? p1 : p2);
          ratio *= std::sqrt(p1 / (p1 + Sv)) / 2 + std::sqrt(p2 / (p2 + Sv)) / 2;
        }
        A = sp(output[1]) * (std::exp(-quad / 2) - ratio);
      }
      EP.values.push_back((Fval)V); EP.advantages.push_back((Fval)((Fval)(V + A) - (Fval)V));
      agent.action = act;
      agent.policyVector = mean; agent.policyVector.insert(agent.policyVector.end(), stdev.begin(), stdev.end());
      // MemoryBuffer::storeAction (MemoryBuffer.cpp:131-170 region): a_t and mu_t next to s_t
      EP.actions.insert(EP.actions.end(), act.begin(), act.end());
      EP.policies.insert(EP.policies.end(), agent.policyVector.begin(), agent.policyVector.end());
    } else {
      // RACER::processTerminal (RACER.cpp:49-59): value of a truncated last state from the network, 0 if terminal
      EP.values.push_back(agent.agentStatus == LAST ? (Fval)scaleNet2V(forward(agent)[0]) : (Fval)0);
      EP.advantages.push_back(0);
      EP.actions.insert(EP.actions.end(), dA, 0.0);                                  // dummy last action / policy
      EP.policies.insert(EP.policies.end(), nOpt ? (size_t)nOpt : 2 * dA, 0.0);
      // MemoryBuffer::terminateCurrentEpisode -> pushBackEpisode
      pushBackEpisode((int)(EP.states.size() / dS), EP.states, EP.actions, EP.policies, EP.rewards, EP.values,
                      agent.agentStatus == TERM, EP.tag, &EP.advantages);
      EP = InProgress();
    }
  }
  bool c_bounded(Uint i) const { return i < MDP.bActionSpaceBounded.size() && MDP.bActionSpaceBounded[i]; }

  // MemoryBuffer::pushBackEpisode: a finished episode enters the training set
  void pushBackEpisode(int nStates, const Fvec& states, const Rvec& actions, const Rvec& policies, const Rvec& rewards,
                       const Fvec& values, bool bReachedTermState, int64_t ID, const Fvec* advantages = nullptr) {
    ck(hl_append_episode(H, nStates, states.data(), actions.data(), policies.data(), rewards.data(), values.data(),
                         advantages ? advantages->data() : nullptr, bReachedTermState ? 1 : 0, ID));
  }

  // the same from the wire format a worker sends (Episode::packEpisode / unpackEpisode, Episode.cpp:24-130)
  void pushBackEpisode(const Fvec& packed) { ck(hl_append_packed_episode(H, packed.data(), (int64_t)packed.size())); }
  // Episode::packEpisode of the stored episode at position `pos` (what MemoryBuffer::save writes per episode)
  Fvec packEpisode(long pos) const {
    int64_t tag; int32_t n, term; ck(hl_get_episode_info(H, pos, &tag, &n, &term));
    Fvec out((Uint)hl_packed_episode_size(H, n));
    ck(hl_pack_episode(H, pos, out.data(), (int64_t)out.size()));
    return out;
  }

  // Learner::initializeLearner (Learner.cpp:47-72)
  void initializeLearner() { ck(hl_initialize(H)); bInit = true; }

  // RACER::setupTasks stepMain + stepComplete, n gradient steps (RACER.cpp:81-108)
  void trainStep(int n = 1) { if (!bInit) die("trainStep before initializeLearner"); ck(hl_step(H, n, nullptr)); }

  long nGradSteps() const { hl_scalars s; ck(hl_get_scalars(H, &s)); return (long)s.nGradSteps; }
  Real beta() const { hl_scalars s; ck(hl_get_scalars(H, &s)); return s.beta; }
  long nStoredSteps() const { hl_scalars s; ck(hl_get_scalars(H, &s)); return (long)s.nStoredSteps; }

  // Learner::getMetrics / getHeaders (Learner.cpp:203-215): the replay-memory columns followed by the network's,
  // formatted as the reference writes them into agent_00_stats.txt (hl_metrics)
  void getMetrics(std::ostringstream& buf) const { char head[1024], line[1024]; ck(hl_metrics(H, head, 1024, line, 1024)); buf << line; }
  void getHeaders(std::ostringstream& buf) const { char head[1024], line[1024]; ck(hl_metrics(H, head, 1024, line, 1024)); buf << head; }
  // Approximator::updateGradStats (Approximator.h:65-68): <learnerName>_net_outGrad_stats.raw, written by the
  // library at the steps with nGradSteps % 1000 == 0 once the name is set; gradStats() = last minibatch
  void setLearnerName(const std::string& learnerName) { ck(hl_set_log_base(H, learnerName.c_str())); }
  void gradStats(std::vector<Real>& mean, std::vector<Real>& rms) const {
    mean.resize((size_t)nOut); rms.resize((size_t)nOut); ck(hl_grad_stats(H, mean.data(), rms.data()));
  }
  // Learner::processStats (Learner.cpp:158-196): one line "<learnID> <step/freqPrint><columns>" appended to
  // <learner>_stats.txt; the header goes to the file once, at the first print
  void processStats(const std::string& learnerName, const bool bPrintHeader, const unsigned freqPrint = 1000, const unsigned learnID = 0) const {
    const unsigned currStep = (unsigned)nGradSteps() + 1, tStamp = currStep / freqPrint;
    std::ostringstream buf, head; getMetrics(buf);
    FILE* fout = std::fopen((learnerName + "_stats.txt").c_str(), "a");
    if (!fout) die("unable to open " + learnerName + "_stats.txt");
    if (bPrintHeader) { getHeaders(head); if (currStep == freqPrint) std::fprintf(fout, "ID #/T   %s\n", head.str().c_str()); }
    std::fprintf(fout, "%02u %05u%s\n", learnID, tStamp, buf.str().c_str());
    std::fclose(fout);
  }

  // Learner::save / restart: the networks ("<base>_net_weights.raw", "_1stMom.raw", "_2ndMom.raw",
  // Approximator.cpp:282-297) and the replay memory ("<base>_scaling.raw", "<base>_rank_RRR_learner_status.raw",
  // "..._data.raw", MemoryBuffer.cpp:172-324)
  void save(const std::string& base, int learnerRank = 0) const {
    ck(hl_save(H, (base + "_net").c_str()));
    ck(hl_save_memory(H, base.c_str(), learnerRank));
  }
  // networks only (a replay memory is restarted only when its files exist, as in the reference)
  void restart(const std::string& base) { ck(hl_restart(H, (base + "_net").c_str())); }
  void restartMemory(const std::string& base, int learnerRank = 0) { ck(hl_restart_memory(H, base.c_str(), learnerRank)); bInit = true; }
};

}  // namespace smarties_amd
