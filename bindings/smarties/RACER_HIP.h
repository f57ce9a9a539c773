// bindings/smarties/RACER_HIP.h -- the class a smarties maintainer adds as Learners/RACER_HIP.h to run the V-RACER / RACER
// learner update on libsmarties_hip.so (include/smarties_hip.h).  It derives from the reference's `Learner`
// (source/smarties/Learners/Learner.h:25-136) and overrides exactly the virtuals the process runtime calls
// (Core/Worker.cpp:150,163-166,214,291-295): selectAction / processTerminal (rollout side), setupTasks (training
// side), initializeLearner, getMetrics / getHeaders / processStats, save / restart.  Everything else of smarties --
// Communicator, Worker / Master, settings, the host MemoryBuffer that collects the episodes in progress
// (Learner::select -> MemoryBuffer::storeState / storeAction / terminateCurrentEpisode, with its
// cumulative_rewards.dat / _obs.raw logs, MemoryBuffer.cpp:479-520) -- is used as it is: finished episodes wait in
// `data->episodes` and are moved to the device at the next turn of the training task.
//
// This file contains no reference source; it compiles only inside the reference tree
// (oracle/Makefile: `make binding` compiles and links it against /root/reference, oracle/binding_check.cpp drives it).
//
// Factory branch (Learners/AlgoFactory.cpp, next to :109, :124, :132-152), e.g. behind a settings key "useHIP":
//   ret = std::make_unique<RACER_HIP<Zero_advantage, Continuous_policy, Rvec>>(MDP, settings, distrib);        // VRACER
//   ret = std::make_unique<RACER_HIP<Param_advantage, Continuous_policy, Rvec>>(MDP, settings, distrib);       // RACER
//   ret = std::make_unique<RACER_HIP<Discrete_advantage, Discrete_policy, Uint>>(MDP, settings, distrib);      // discrete
#pragma once
#include "Learner.h"
#include "../Utils/FunctionUtilities.h"
#include "../Math/Continuous_policy.h"
#include "../Math/Discrete_policy.h"
#include "../Math/Zero_advantage.h"
#include "../Math/Gaus_advantage.h"
#include "../Math/Discrete_advantage.h"
#include "smarties_hip.h"

#include <mutex>
#include <sstream>
#include <type_traits>

namespace smarties
{

template<typename Advantage_t, typename Policy_t, typename Action_t>
class RACER_HIP : public Learner
{
  static constexpr bool bDiscrete = std::is_same<Action_t, Uint>::value;
  static constexpr bool bGaussAdv = std::is_same<Advantage_t, Gaussian_advantage>::value;

  hl_learner* H = nullptr;
  const Uint nA = Policy_t::compute_nA(aInfo), nL = Advantage_t::compute_nL(aInfo);
  // output groups as RACER<...>::count_outputs / count_pol_starts / count_adv_starts lay them out
  // (Learners/RACER_common.cpp:117-186): [V | advantage (nL) | policy (nA) | stdev parameters (nA, continuous)]
  const std::vector<Uint> pol_start = bDiscrete ? std::vector<Uint>{1 + nL} : std::vector<Uint>{1 + nL, 1 + nL + nA};
  const std::vector<Uint> adv_start = std::vector<Uint>{1};
  const Uint VsID = 0;
  Uint nOutputs = 0, nInputs = 0;
  bool bRecurrent = false;

  void ck(const int rc) const { if (rc) die(hl_last_error(H)); }                 // the reference's convention: die()

  static int hlFunc(const std::string& f) {                                       // makeFunction (Functions.h:643-668)
    const char* names[] = {"Linear", "Tanh", "SoftSign", "Relu", "LRelu", "Sigm", "HardSign", "SoftPlus", "ExpPlus", "Exp"};
    for (int i = 0; i < 10; ++i) if (f == names[i]) return i;
    die("Activation function not recognized"); return 0;
  }

  // Approximator::forward(agent) (Network/Approximator.h:300-330): the network outputs for the state the agent is in.  The
  // library standardises raw states itself; with appended observations it gets the state of step t followed by those
  // of t-1 .. t-nAppendedObs (Episode::standardizedState, Episode.h:172-183), recurrent layers the last steps of the episode.
  Rvec forward(const MiniBatch& MB) const
  {
    const Episode& EP = MB.getEpisode(0);
    const Uint t = EP.nsteps() - 1, dS = MDP.dimStateObserved;
    Rvec output(nOutputs);
    if (bRecurrent) {
      const Uint n = std::min((Uint) settings.nnBPTTseq, t) + 1;
      std::vector<float> S(n * dS);
      for (Uint k = 0; k < n; ++k) std::copy(EP.states[t + 1 - n + k].begin(), EP.states[t + 1 - n + k].end(), S.begin() + k * dS);
      ck(hl_forward_sequence(H, (int32_t) n, S.data(), output.data()));
    } else {
      std::vector<float> S((1 + MDP.nAppendedObs) * dS);
      for (Uint j = 0; j <= MDP.nAppendedObs; ++j) {
        const Uint tt = t >= j ? t - j : 0;
        std::copy(EP.states[tt].begin(), EP.states[tt].end(), S.begin() + j * dS);
      }
      ck(hl_forward(H, 1, S.data(), output.data()));
    }
    return output;
  }

  // finished episodes collected by the host MemoryBuffer move to the device (MemoryBuffer::pushBackEpisode is where the
  // reference makes them visible to the sampler); the host copies are dropped, the host counters follow the device's
  void moveEpisodesToDevice()
  {
    std::lock_guard<std::mutex> lock(data->dataset_mutex);
    const Uint dS = MDP.dimStateObserved, dA = aInfo.dim(), dP = MDP.policyVecDim;
    for (auto & ep : data->episodes) {
      const Episode& EP = * ep;
      const Uint N = EP.nsteps();
      std::vector<float> S(N * dS), V(N), ADV(N);
      std::vector<double> A(N * dA, 0), P(N * dP, 0), R(N);
      for (Uint t = 0; t < N; ++t) {
        std::copy(EP.states[t].begin(), EP.states[t].end(), S.begin() + t * dS);
        if (EP.actions[t].size() == dA)  std::copy(EP.actions[t].begin(),  EP.actions[t].end(),  A.begin() + t * dA);
        if (EP.policies[t].size() == dP) std::copy(EP.policies[t].begin(), EP.policies[t].end(), P.begin() + t * dP);
        R[t] = EP.rewards[t]; V[t] = EP.stateValue[t]; ADV[t] = EP.actionAdvantage[t];
      }
      ck(hl_append_episode(H, (int32_t) N, S.data(), A.data(), P.data(), R.data(), V.data(), ADV.data(),
                           EP.bReachedTermState, (int64_t) EP.ID));
    }
    data->episodes.clear();
    int64_t nObs = 0, nEps = 0;
    ck(hl_get_counts(H, &nObs, &nEps, nullptr, nullptr, nullptr));
    data->counters.nTransitions = nObs; data->counters.nEpisodes = 0;      // (the host vector is empty: pushBackEpisode's own invariant)
    nStoredOnDevice = nObs;
  }
  long nStoredOnDevice = 0;

 public:
  RACER_HIP(MDPdescriptor& M, HyperParameters& S, ExecutionInfo& D) : Learner(M, S, D)
  {
    hl_config c{}; c.struct_size = sizeof(c);
    c.dimS = (int32_t) M.dimStateObserved; c.dimA = (int32_t) aInfo.dim();
    for (Uint i = 0; i < aInfo.dim() && !bDiscrete; ++i) c.bounded[i] = aInfo.isBounded(i);
    c.n_hidden = (int32_t) S.nnLayerSizes.size();
    // Learner_approximator::createEncoder (:149-166): the encoder's dense layers head the same network (zero entries dropped there)
    c.n_encoder = (int32_t) S.encoderLayerSizes.size();
    if (c.n_hidden + c.n_encoder > HL_MAX_HIDDEN) die("too many hidden layers for hl_config");
    for (Uint i = 0; i < S.nnLayerSizes.size(); ++i) c.hidden[i] = (int32_t) S.nnLayerSizes[i];
    for (Uint i = 0; i < S.encoderLayerSizes.size(); ++i) c.encoder[i] = (int32_t) S.encoderLayerSizes[i];
    c.nnFunc = hlFunc(S.nnFunc);
    c.nnOutputFunc = hlFunc(S.nnOutputFunc);                                      // Approximator.cpp:193,228
    // what Approximator::buildFromSettings makes of nnType (Approximator.cpp:218-226)
    const std::string netType = M.isPartiallyObservable && S.bRecurrent == false ? "MGU" : S.nnType;
    c.nn_type = netType == "LSTM" ? HL_NN_LSTM : (netType == "MGU" || netType == "GRU") ? HL_NN_MGU :
                netType == "RNN" ? HL_NN_RNN : (netType == "FFNN" ? HL_NN_FFNN : -1);
    // settings the library does not serve die here rather than train something else (the reference's convention):
    //  * any other nnType string makes plain dense layers in Builder::addLayer, "Recurrent" recurrent ones without a BPTT
    //    window (HyperParameters.cpp:209 does not count it as recurrent);
    if (c.nn_type < 0) die("nnType not served by the HIP library (FFNN, LSTM, MGU / GRU, RNN)");
    // a partially observable MDP with nnType left non-recurrent gets "RNN" encoder layers under its "MGU" layers
    // (Approximator.cpp:264-270 vs :221-223)
    c.encoder_rnn = (M.isPartiallyObservable && S.bRecurrent == false) ? 1 : 0;
    bRecurrent = c.nn_type != HL_NN_FFNN; c.nnBPTTseq = (int32_t) S.nnBPTTseq;
    // MemoryProcessing::createReturnEstimator (:418-450); AlgoFactory.cpp:134-135 turns "default" into "retrace" for this learner
    c.returnsEstimator = (S.returnsEstimator == "default" || S.returnsEstimator == "retrace") ? HL_RET_RETRACE :
                         S.returnsEstimator == "retraceExplore" ? HL_RET_RETRACE_EXPLORE : S.returnsEstimator == "GAE" ? HL_RET_GAE :
                         S.returnsEstimator == "none" ? HL_RET_NONE : HL_RET_RETRACE /* createReturnEstimator's own fall-through */;
    c.adv_kind = bDiscrete ? HL_ADV_DISCRETE : (bGaussAdv ? HL_ADV_GAUSSIAN : HL_ADV_ZERO);
    c.n_options = bDiscrete ? (int32_t) nA : 0;
    // removal rule of an over-full replay and minibatch sampler (getERfilterAlgo, MemoryProcessing.cpp:261-298;
    // Sampling::prepareSampler, Sampling.cpp:298-340)
    c.ERoldSeqFilter = S.ERoldSeqFilter == "oldest" ? HL_ER_OLDEST : S.ERoldSeqFilter == "farpolfrac" ? HL_ER_FARPOLFRAC :
                       S.ERoldSeqFilter == "maxkldiv" ? HL_ER_MAXKLDIV : S.ERoldSeqFilter == "minerror" ? HL_ER_MINERROR : -1;
    if (c.ERoldSeqFilter < 0) die("ERoldSeqFilter setting not recognized.");
    c.dataSamplingAlgo = S.dataSamplingAlgo == "uniform" ? HL_SAMPLE_UNIFORM : S.dataSamplingAlgo == "PERrank" ? HL_SAMPLE_PERRANK :
                         S.dataSamplingAlgo == "PERerr" ? HL_SAMPLE_PERERR : S.dataSamplingAlgo == "PERseq" ? HL_SAMPLE_PERSEQ : -1;
    if (c.dataSamplingAlgo < 0) die("Setting dataSamplingAlgo not recognized.");
    c.nAppendedObs = (int32_t) M.nAppendedObs;
    c.n_conv = (int32_t) M.conv2dDescriptors.size();
    if (c.n_conv > HL_MAX_CONV) die("too many convolutional layers for hl_config");
    for (int j = 0; j < c.n_conv; ++j) {
      const Conv2D_Descriptor& d = M.conv2dDescriptors[j]; hl_conv2d& o = c.conv[j];
      o.inpFeatures = d.inpFeatures; o.inpY = d.inpY; o.inpX = d.inpX; o.outFeatures = d.outFeatures; o.outY = d.outY; o.outX = d.outX;
      o.filterx = d.filterx; o.filtery = d.filtery; o.stridex = d.stridex; o.stridey = d.stridey; o.paddinx = d.paddinx; o.paddiny = d.paddiny;
    }
    c.batchSize = (int32_t) S.batchSize; c.maxTotObsNum = (int64_t) S.maxTotObsNum; c.minTotObsNum = (int64_t) S.minTotObsNum;   // GLOBAL:
    c.gamma = S.gamma; c.lambda = S.lambda; c.clipImpWeight = S.clipImpWeight; c.penalTol = S.penalTol;   // the library splits them over
    c.epsAnneal = S.epsAnneal; c.learnrate = S.learnrate; c.nnLambda = S.nnLambda; c.explNoise = S.explNoise;   // the replicas like
    c.outWeightsPrefac = S.outWeightsPrefac; c.randSeed = D.randSeed;                                       // HyperParameters.cpp:186-197
    c.n_ranks = (int32_t) learn_size; c.rank = (int32_t) learn_rank; c.device_id = -1;                       // rank % device count
    c.ref_threads = (int32_t) D.nThreads;
    const int rc = hl_create(&c, &H);
    if (rc) { const std::string msg = H ? hl_last_error(H) : hl_status_string(rc); if (H) hl_destroy(H); H = nullptr; die(msg.c_str()); }
    ck(hl_init_weights(H));
    if (learn_size > 1 && learn_size <= 16 && getenv("SMARTIES_HIP_RCCL") == nullptr) {
      // learners of one node: every replica's exchange window is mapped by its peers (xGMI); the 96-byte handles travel over
      // the learners' MPI communicator (Optimizer.h:24), the sums themselves never touch MPI or the host
      uint8_t mine[HL_XCHG_HANDLE_BYTES];
      std::vector<uint8_t> all((size_t) learn_size * HL_XCHG_HANDLE_BYTES);
      ck(hl_xchg_export(H, mine));
      MPI_Allgather(mine, HL_XCHG_HANDLE_BYTES, MPI_BYTE, all.data(), HL_XCHG_HANDLE_BYTES, MPI_BYTE, learnersComm);
      ck(hl_xchg_connect(H, all.data()));
    } else if (learn_size > 1) {                       // RCCL id over the same communicator
      uint8_t id[128] = {0};
      if (learn_rank == 0) ck(hl_comm_unique_id(id));
      MPI_Bcast(id, 128, MPI_BYTE, 0, learnersComm);
      ck(hl_comm_init(H, id));
    }
    nOutputs = (Uint) hl_num_outputs(H);
    nInputs = (1 + M.nAppendedObs) * M.dimStateObserved;
  }
  ~RACER_HIP() override { if (H) hl_destroy(H); }

  static Uint getnDimPolicy(const ActionInfo& aI) { return bDiscrete ? aI.dimDiscrete() : 2 * aI.dim(); }

  // RACER::selectAction (Learners/RACER.cpp:30-47)
  void selectAction(const MiniBatch& MB, Agent& agent) override
  {
    const Rvec output = forward(MB);
    const Policy_t pol(pol_start, aInfo, output);
    auto action = pol.selectAction(agent, distrib.bTrain);
    const Advantage_t adv(adv_start, aInfo, output, &pol);
    const Real V = scaleNet2V(output[VsID]);
    MB.appendValues(V, V + adv.computeAdvantage(action));
    agent.setAction(action, pol.getVector());
  }
  // RACER::processTerminal (:49-59)
  void processTerminal(const MiniBatch& MB, Agent& agent) override
  {
    if (agent.agentStatus == LAST) MB.appendValues(scaleNet2V(forward(MB)[VsID]));
    else MB.appendValues(0);
  }
  static Real scaleNet2V(const Real x) {                               // Learners/RACER_common.cpp:23-27
    return x > 0 ? 100 * (x + 51) - 100 * std::sqrt(2601 + 100 * x) : 100 * (x - 51) + 100 * std::sqrt(2601 - 100 * x);
  }

  // Learner::initializeLearner (Learner.cpp:47-72)
  void initializeLearner() override
  {
    moveEpisodesToDevice();
    // a restarted learner skips the start-up passes (Learner.cpp:51-54) -- Worker calls setupTasks() right after restart()
    // (Core/Worker.cpp:293-294), which sends the task queue through here once more; hl_restart_memory left the library ready to step
    if (nGradSteps() > 0) return;
    ck(hl_initialize(H));
    data->counters.nGatheredB4Startup = nObsB4StartTraining;
  }

  // RACER::setupTasks (RACER.cpp:62-110): the gating stays the reference's, stepMain + stepComplete are one hl_step
  void setupTasks(TaskQueue& tasks) override
  {
    if (not bTrain) return;
    algoSubStepID = -1;
    tasks.add([&]() {
      if (algoSubStepID >= 0) return;
      if (data->nStoredSteps() < nObsB4StartTraining) return;
      initializeLearner();
      algoSubStepID = 0;
    });
    tasks.add([&]() {
      if (algoSubStepID not_eq 0) return;
      if (blockGradientUpdates()) return;
      moveEpisodesToDevice();
      ck(hl_step(H, 1, nullptr));             // sample .. Adam .. ReF-ER bookkeeping; returns when the work is queued
      logStats();
      globalGradCounterUpdate();
    });
  }

  // the line of <learner>_stats.txt: hl_metrics holds the replay columns AND the network's (Learner::processStats prints
  // data->getMetrics first: those columns come from the device here)
  void processStats(const bool bPrintHeader) override
  {
    const unsigned currStep = nGradSteps() + 1, tStamp = currStep / freqPrint;
    char head[2048], line[2048];
    ck(hl_metrics(H, head, 2048, line, 2048));
    if (learn_rank) return;
    FILE* fout = fopen((learner_name + "_stats.txt").c_str(), "a");
    if (bPrintHeader) {
      printf("ID #/T   %s\n", head);
      if (currStep == freqPrint) fprintf(fout, "ID #/T   %s\n", head);
    }
    printf("%02u %05u%s\n", (unsigned) data->learnID, tStamp, line);
    fprintf(fout, "%02u %05u%s\n", (unsigned) data->learnID, tStamp, line);
    fclose(fout); fflush(0);
  }
  void getMetrics(std::ostringstream& buff) const override { char head[2048], line[2048]; ck(hl_metrics(H, head, 2048, line, 2048)); buff << line; }
  void getHeaders(std::ostringstream& buff) const override { char head[2048], line[2048]; ck(hl_metrics(H, head, 2048, line, 2048)); buff << head; }

  // Learner_approximator::save / restart + Learner::save / restart: the same files
  // (<name>_net_weights.raw ..., <name>_scaling.raw, <name>_rank_RRR_learner_{status,data}.raw)
  void save() override
  {
    ck(hl_save(H, (learner_name + "_net").c_str()));
    ck(hl_save_memory(H, learner_name.c_str(), (int32_t) learn_rank));
  }
  void restart() override
  {
    if (distrib.restart == "none") return;
    const std::string base = distrib.restart + "/" + learner_name;
    ck(hl_restart(H, (base + "_net").c_str()));
    if (hl_restart_memory(H, base.c_str(), (int32_t) learn_rank) == HL_OK) {
      int64_t nObs = 0, nEps = 0, nGrad = 0, seenS = 0, seenE = 0;
      ck(hl_get_counts(H, &nObs, &nEps, &nGrad, &seenS, &seenE));
      data->counters.nTransitions = nObs; data->counters.nGradSteps = nGrad;
      data->counters.nSeenTransitions_loc = seenS; data->counters.nSeenEpisodes_loc = seenE;
      int64_t nInit = 0; ck(hl_get_initial_data(H, &nInit));
      data->counters.nGatheredB4Startup = (long) nInit;      // MemoryBuffer::restart (:249-250): blockGradientUpdates counts from here
      algoSubStepID = 0;                     // a restarted learner skips initializeLearner (Learner.cpp:51-54)
    }
  }

  hl_learner* handle() const { return H; }
};

} // end namespace smarties
