/*
 * oracle/binding_check.cpp -- compiles bindings/smarties/RACER_HIP.h inside the reference tree and links it with the
 * reference's own objects and libsmarties_hip.so (oracle/Makefile: `make binding`; build container only, the binary travels
 * to the GPU box under oracle/_ref/).  TEST INFRASTRUCTURE: contains no reference source.
 *
 * Run on a GPU box it drives the class the way Core/Worker.cpp does: agents go through Learner::select (the plug-in path:
 * MemoryBuffer::storeState / selectAction / storeAction / terminateCurrentEpisode), the TaskQueue runs setupTasks'
 * lambdas, and it prints what happened as one JSON line.  The same episodes are fed to the reference's own
 * RACER<Zero_advantage, Continuous_policy, Rvec> in the same process, so that the two learners can be compared on
 * identical data: same number of stored transitions, same weights after initialisation, beta / weight norm after n steps
 * within the sampling noise (the two draw their minibatches in different episode orders).
 */
#include <sstream>
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
#include "smarties/Utils/TaskQueue.h"
#include "smarties/Learners/RACER_HIP.h"          // -> bindings/smarties/RACER_HIP.h through the include path of the recipe
#undef protected
#undef private

#include <cstdio>
#include <random>

using namespace smarties;

int main(int argc, char** argv)
{
  const long nSteps = argc > 1 ? atol(argv[1]) : 50;
  int one = 1; char* av[] = {argv[0], nullptr}; char** avp = av;
  ExecutionInfo info(one, avp);
  info.nThreads = 1; omp_set_num_threads(1);
  info.randSeed = 42; info.initialze();
  info.learners_train_comm = MPI_COMM_SELF; info.bIsMaster = true;
  info.nAgents = 1; info.nOwnedEnvironments = 1; info.nEnvironments = 1;
  info.logAllSamples = 0; info.learnersOnWorkers = false; info.restart = "none";
  const Uint dS = 17, dA = 6;
  auto makeMDP = [&](MDPdescriptor& M) {
    M.dimState = dS; M.dimAction = dA;
    M.bActionSpaceBounded = std::vector<bool>(dA, true);
    M.synchronize([](void*, size_t) {});
    M.policyVecDim = 2 * dA;
  };
  MDPdescriptor MDP1, MDP2; makeMDP(MDP1); makeMDP(MDP2);
  auto makeHP = [&]() {
    auto HP = std::make_unique<HyperParameters>(dS, dA);
    HP->learner = "VRACER"; HP->returnsEstimator = "retrace";
    HP->nnLayerSizes = {64, 64}; HP->nnFunc = "SoftSign"; HP->batchSize = 32;
    HP->maxTotObsNum = 4096; HP->minTotObsNum = 1024; HP->obsPerStep = 0; HP->saveFreq = 1000000000;
    HP->defineDistributedLearning(info); HP->check();
    return HP;
  };
  auto HP1 = makeHP(), HP2 = makeHP();
  using REF = RACER<Zero_advantage, Continuous_policy, Rvec>;
  using HIP = RACER_HIP<Zero_advantage, Continuous_policy, Rvec>;
  // same generator state for both weight initialisations: the reference draws from generators[0] at construction
  const std::mt19937 g0 = info.generators[0];
  auto Lref = std::make_unique<REF>(MDP1, *HP1, info);
  info.generators[0] = g0;
  std::unique_ptr<Learner> Lhip = std::make_unique<HIP>(MDP2, *HP2, info);      // through the base class, as createLearner returns it
  Lref->setLearnerName("ref_00", 0); Lhip->setLearnerName("hip_00", 1);
  TaskQueue qRef([]() { return false; }), qHip([]() { return false; });
  Lref->setupTasks(qRef); Lhip->setupTasks(qHip);

  // the plug-in path: one agent per learner, identical observations; each learner picks its own actions
  Agent A1(0, 0, 0, MDP1), A2(0, 0, 0, MDP2);
  A1.initializeActionSampling(info.generators[0]); A2.initializeActionSampling(info.generators[0]);
  std::mt19937 g(7); std::normal_distribution<double> N01(0, 1);
  long nEp = 0;
  while (Lref->locDataSetSize() < 1500) {
    std::vector<double> s(dS);
    for (auto& x : s) x = N01(g);
    A1.update(INIT, s, 0.0); A2.update(INIT, s, 0.0); Lref->select(A1); Lhip->select(A2);
    const int T = 20 + (int)(nEp % 30);
    for (int t = 1; t <= T; ++t) {
      for (auto& x : s) x = N01(g);
      const double r = N01(g);
      const episodeStatus st = t == T ? (nEp % 3 ? LAST : TERM) : CONT;
      A1.update(st, s, r); A2.update(st, s, r); Lref->select(A1); Lhip->select(A2);
    }
    ++nEp;
  }
  const long storedRef = Lref->locDataSetSize(), storedHipHost = Lhip->locDataSetSize();
  qRef.run(); qHip.run();                                   // stepInit (+ first step)
  while (Lref->nGradSteps() < nSteps) qRef.run();
  while (Lhip->nGradSteps() < nSteps) qHip.run();
  HIP* hp = dynamic_cast<HIP*>(Lhip.get());
  hl_scalars sc; hl_get_scalars(hp->handle(), &sc);
  std::vector<float> w((size_t) hl_num_params(hp->handle()));
  hl_get_params(hp->handle(), w.data(), nullptr, nullptr);
  long double wn = 0; for (float x : w) wn += (long double) x * x;
  const long double wnRef = Lref->networks[0]->net->weights->compute_weight_norm();
  std::ostringstream mh; Lhip->getMetrics(mh);
  printf("{\"episodes\": %ld, \"stored_ref\": %ld, \"stored_hip_host\": %ld, \"stored_hip_device\": %ld, \"steps_ref\": %ld, \"steps_hip\": %ld, "
         "\"beta_ref\": %.9g, \"beta_hip\": %.9g, \"wnorm_ref\": %.9Lg, \"wnorm_hip\": %.9Lg, \"nparams_ref\": %lu, \"nparams_hip\": %lu}\n",
         nEp, storedRef, storedHipHost, (long) sc.nStoredSteps, Lref->nGradSteps(), Lhip->nGradSteps(), (double) Lref->data->beta, sc.beta,
         wnRef, std::sqrt(wn), (unsigned long) Lref->networks[0]->net->weights->nParams, (unsigned long) w.size());
  printf("hip stats line:%s\n", mh.str().c_str());
  if (argc > 2 && std::string(argv[2]) == "restart") {
    // Both learners write their checkpoints (Learner_approximator::save -> <name>_net_*.raw, <name>_scaling.raw,
    // <name>_rank_000_learner_{status,data}.raw) and two NEW learners start from them the way Core/Worker.cpp:291-295 does:
    // restart(), then setupTasks() -- whose first task runs initializeLearner() again (a restarted learner must skip it,
    // Learner.cpp:51-54) --, then train on.  The binding reads the REFERENCE's files as well as its own.
    const long more = argc > 3 ? atol(argv[3]) : 100;
    Lref->save(); Lhip->save();
    hl_scalars before; hl_get_scalars(hp->handle(), &before);
    info.restart = ".";
    MDPdescriptor MDP3, MDP4, MDP5; makeMDP(MDP3); makeMDP(MDP4); makeMDP(MDP5);
    auto HP3 = makeHP(), HP4 = makeHP(), HP5 = makeHP();
    auto Rref = std::make_unique<REF>(MDP3, *HP3, info);
    std::unique_ptr<Learner> Rhip = std::make_unique<HIP>(MDP4, *HP4, info);      // restarts from its own files
    std::unique_ptr<Learner> Xhip = std::make_unique<HIP>(MDP5, *HP5, info);      // restarts from the reference's files
    Rref->setLearnerName("ref_00", 0); Rhip->setLearnerName("hip_00", 1); Xhip->setLearnerName("ref_00", 2);
    TaskQueue q3([]() { return false; }), q4([]() { return false; }), q5([]() { return false; });
    Rref->restart(); Rref->setupTasks(q3);
    Rhip->restart(); Rhip->setupTasks(q4);
    Xhip->restart(); Xhip->setupTasks(q5);
    HIP* rp = dynamic_cast<HIP*>(Rhip.get()); HIP* xp = dynamic_cast<HIP*>(Xhip.get());
    hl_scalars r0, x0; hl_get_scalars(rp->handle(), &r0); hl_get_scalars(xp->handle(), &x0);
    const long g0r = Rref->nGradSteps(), g0h = Rhip->nGradSteps(), g0x = Xhip->nGradSteps();
    Rref->initializeLearner(); Rhip->initializeLearner(); Xhip->initializeLearner();      // what the first task does: no start-up passes for restarted learners
    hl_scalars r1, x1; hl_get_scalars(rp->handle(), &r1); hl_get_scalars(xp->handle(), &x1);
    while (Rref->nGradSteps() < g0r + more) q3.run();
    while (Rhip->nGradSteps() < g0h + more) q4.run();
    while (Xhip->nGradSteps() < g0x + more) q5.run();
    hl_scalars r2, x2; hl_get_scalars(rp->handle(), &r2); hl_get_scalars(xp->handle(), &x2);
    auto wnormOf = [](hl_learner* H) { std::vector<float> v((size_t) hl_num_params(H)); hl_get_params(H, v.data(), nullptr, nullptr);
                                       long double a = 0; for (float x : v) a += (long double) x * x; return (double) std::sqrt(a); };
    printf("{\"restart\": 1, \"grad0_ref\": %ld, \"grad0_hip\": %ld, \"grad0_x\": %ld, \"beta_saved\": %.12g, \"beta_restarted\": %.12g, "
           "\"beta_after_init_task\": %.12g, \"beta_x_restarted\": %.12g, \"beta_x_after_init_task\": %.12g, \"beta_ref_saved\": %.12g, "
           "\"stored_saved\": %ld, \"stored_restarted\": %ld, \"stored_x\": %ld, \"stored_ref\": %ld, "
           "\"steps_ref\": %ld, \"steps_hip\": %ld, \"steps_x\": %ld, \"beta_ref\": %.9g, \"beta_hip\": %.9g, \"beta_x\": %.9g, "
           "\"wnorm_ref\": %.9g, \"wnorm_hip\": %.9g, \"wnorm_x\": %.9g}\n",
           g0r, g0h, g0x, before.beta, r0.beta, r1.beta, x0.beta, x1.beta, (double) Lref->data->beta,
           (long) before.nStoredSteps, (long) r0.nStoredSteps, (long) x0.nStoredSteps, (long) Rref->locDataSetSize(),
           Rref->nGradSteps(), Rhip->nGradSteps(), Xhip->nGradSteps(), (double) Rref->data->beta, r2.beta, x2.beta,
           (double) Rref->networks[0]->net->weights->compute_weight_norm(), wnormOf(rp->handle()), wnormOf(xp->handle()));
  }
  return 0;
}
