// tests/cpp/host_parity.cpp -- C++ parity test of the host side (smarties_amd/host/vracer_hip.h) against
// the CPU oracle (oracle/port, ol_* C API).  Reads like a smarties learner test: build the MDP and the
// settings, create the learner, push episodes, initializeLearner, train, compare.
// Built by __graft_entry__.build(); run on an MI355X by tests/test_host_cpp.py (-m gpu).
#include <cassert>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "../../smarties_amd/host/vracer_hip.h"
#include "../../oracle/port/vracer_port.h"   // test infrastructure: the checker
#include "../../oracle/synth.h"

using namespace smarties_amd;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static double relinf(const std::vector<float>& a, const std::vector<float>& b) {
  double num = 0, den = 1e-30;
  for (size_t i = 0; i < a.size(); ++i) { num = std::max(num, (double)std::fabs(a[i] - b[i])); den = std::max(den, (double)std::fabs(b[i])); }
  return num / den;
}

int main() {
  // ---- problem: the north-star shape on a small replay -------------------------------------------------
  MDPdescriptor MDP; MDP.dimStateObserved = 17; MDP.dimAction = 6; MDP.bActionSpaceBounded.assign(6, true);
  HyperParameters HP; HP.nnLayerSizes = {256, 256}; HP.nnFunc = "SoftSign"; HP.batchSize = 64; HP.maxTotObsNum = 20000;
  HP.clipImpWeight = 4; HP.epsAnneal = 0; HP.explNoise = 0.4472135955; HP.outWeightsPrefac = 0.1; HP.randSeed = 42;
  VRACER L(MDP, HP, /*deviceID*/0);

  hl_config c{}; c.struct_size = sizeof(c); c.dimS = 17; c.dimA = 6; for (int i = 0; i < 6; ++i) c.bounded[i] = 1;
  c.n_hidden = 2; c.hidden[0] = c.hidden[1] = 256; c.nnFunc = HL_FUNC_SOFTSIGN; c.adv_kind = HL_ADV_ZERO;
  c.batchSize = 64; c.maxTotObsNum = 20000; c.gamma = HP.gamma; c.lambda = HP.lambda; c.clipImpWeight = 4; c.penalTol = HP.penalTol;
  c.epsAnneal = 0; c.learnrate = HP.learnrate; c.nnLambda = HP.nnLambda; c.explNoise = HP.explNoise; c.outWeightsPrefac = 0.1;
  c.randSeed = 42; c.n_ranks = 1; c.ref_threads = 1;
  ol_learner* O = nullptr;
  CHECK(ol_create(&c, &O) == 0, "ol_create");
  CHECK(ol_init_weights(O) == 0, "ol_init_weights");
  const int64_t nP = hl_num_params(L.handle());
  std::vector<float> wG(nP), wO(nP), m1(nP), m2(nP);
  hl_get_params(L.handle(), wG.data(), m1.data(), m2.data()); ol_get_params(O, wO.data(), m1.data(), m2.data());
  CHECK(wG == wO, "initial weights differ (Layer::initialize draw order)");

  // ---- replay: the same synthetic episodes into both ----------------------------------------------------
  synth_cfg sc{7, 17, 6, 30, 90, 0.3, 0.5, 1.0};
  for (uint64_t e = 0; e < 80; ++e) {
    int term = 0; const int N = synth_episode_len(&sc, e, &term);
    Fvec S((size_t)N * 17), V(N); Rvec A((size_t)N * 6), MU((size_t)N * 12), R(N);
    synth_episode(&sc, e, S.data(), A.data(), MU.data(), R.data(), V.data());
    L.pushBackEpisode(N, S, A, MU, R, V, term != 0, (int64_t)e);
    CHECK(ol_append_episode(O, N, S.data(), A.data(), MU.data(), R.data(), V.data(), nullptr, term, (int64_t)e) == 0, "ol_append_episode");
  }
  bool threw = false;
  try { L.trainStep(1); } catch (const std::runtime_error&) { threw = true; }
  CHECK(threw, "trainStep before initializeLearner must die");
  L.initializeLearner(); CHECK(ol_initialize(O) == 0, "ol_initialize");

  // ---- training: sampled indices bit-exact, weights / beta within the fp32 tolerance ---------------------
  const int B = 64;
  std::vector<int64_t> fG(B), fO(B);
  for (int k = 1; k <= 30; ++k) {
    L.trainStep(1); CHECK(ol_step(O, 1, nullptr) == 0, "ol_step");
    hl_readback(L.handle(), HL_TAP_FLAT, fG.data(), B * 8); ol_readback(O, HL_TAP_FLAT, fO.data(), B * 8);
    CHECK(fG == fO, "step %d: sampled indices differ", k);
  }
  L.trainStep(170); CHECK(ol_step(O, 170, nullptr) == 0, "ol_step(170)");   // replayed graphs on the device side
  hl_get_params(L.handle(), wG.data(), m1.data(), m2.data()); ol_get_params(O, wO.data(), m1.data(), m2.data());
  CHECK(relinf(wG, wO) < 1e-4, "weights after 200 steps: rel err %.3g", relinf(wG, wO));
  hl_scalars sO; ol_get_scalars(O, &sO);
  CHECK(std::fabs(L.beta() - sO.beta) <= 1e-9 * sO.beta, "beta %.17g vs %.17g", L.beta(), sO.beta);
  CHECK(L.nGradSteps() == 200, "nGradSteps %ld", L.nGradSteps());
  uint32_t rG[625], rO[625]; hl_get_rng_state(L.handle(), rG); ol_get_rng_state(O, rO);
  CHECK(std::memcmp(rG, rO, sizeof(rG)) == 0, "generator state differs after 200 steps");

  // ---- acting: Learner::select on a live agent -----------------------------------------------------------
  {
    Agent agent(0, 123);
    const long before = L.nStoredSteps();
    const int T = 12;
    for (int t = 0; t <= T; ++t) {
      agent.agentStatus = t == 0 ? INIT : (t == T ? LAST : CONT);
      agent.state.resize(17); for (int i = 0; i < 17; ++i) agent.state[i] = 0.1f * (float)std::sin(0.7 * t + i);
      agent.reward = 0.5 + 0.01 * t;
      std::mt19937 genCopy = agent.generator;
      L.select(agent);
      if (t < T) {
        std::vector<double> out(13);
        CHECK(ol_forward(O, 1, agent.state.data(), out.data()) == 0, "ol_forward");
        for (int i = 0; i < 6; ++i) {
          const double p = out[7 + i], sd = (p + std::sqrt(1 + p * p)) / 2;
          CHECK(std::fabs(agent.policyVector[i] - out[1 + i]) <= 1e-5 * (1 + std::fabs(out[1 + i])), "policy mean %d at t=%d", i, t);
          CHECK(std::fabs(agent.policyVector[6 + i] - sd) <= 1e-12 * sd, "policy stdev %d", i);
          double a = agent.policyVector[i] + agent.policyVector[6 + i] * VRACER::sampleClippedGaussian(genCopy);
          a = std::min(8.31776613503286, std::max(-8.31776613503286, a));
          CHECK(agent.action[i] == a, "action %d at t=%d: generator draw order", i, t);
        }
      }
    }
    CHECK(L.nStoredSteps() == before + T, "episode of %d transitions entered the replay (%ld -> %ld)", T, before, L.nStoredSteps());
    L.trainStep(3);                                   // the appended episode is sampled from now on
    CHECK(L.nGradSteps() == 203, "nGradSteps after appended episode");
  }

  // ---- wire format: a stored episode packed and pushed into a second learner --------------------------------
  {
    const Fvec packed = L.packEpisode(3);
    VRACER L3(MDP, HP, 0);
    L3.pushBackEpisode(packed);
    L3.pushBackEpisode(L.packEpisode(5));
    L3.initializeLearner();
    const Fvec again = L3.packEpisode(1);          // newest first: position 1 is the first one pushed
    const size_t nfl = packed.size() - 10, N = nfl / (17 + 1 + 6 + 12 + 6), tup = 17 + 1 + 6 + 12;
    CHECK(again.size() == packed.size(), "packed size");
    CHECK(std::memcmp(again.data(), packed.data(), N * tup * sizeof(float)) == 0, "states / rewards / actions / policies survive the wire format");
  }

  // ---- checkpoint round trip through the reference's file format --------------------------------------------
  {
    char tmpl[] = "/tmp/smarties_hip_XXXXXX"; const char* dir = mkdtemp(tmpl);
    CHECK(dir != nullptr, "mkdtemp");
    const std::string base = std::string(dir) + "/agent_00";
    L.save(base);
    VRACER L2(MDP, HP, 0);
    L2.restart(base);
    std::vector<float> w2(nP), a1(nP), a2(nP), b1(nP), b2(nP);
    hl_get_params(L.handle(), wG.data(), a1.data(), a2.data()); hl_get_params(L2.handle(), w2.data(), b1.data(), b2.data());
    CHECK(wG == w2 && a1 == b1 && a2 == b2, "checkpoint round trip");
    bool missing = false;
    try { L2.restart(std::string(dir) + "/nothing_here"); } catch (const std::runtime_error&) { missing = true; }
    CHECK(missing, "restart from a missing file must die");
    std::ostringstream m, hd; L.getMetrics(m); L.getHeaders(hd);
    CHECK(hd.str().rfind("|  avgR  | avgr | stdr | DKL ", 0) == 0 && hd.str().find("| net") != std::string::npos, "getHeaders");
    {   // one number per header column
      std::istringstream is(m.str()); double v; int nNum = 0; while (is >> v) ++nNum;
      int nCol = 0; for (char c : hd.str()) nCol += c == '|';
      CHECK(nNum == nCol && nNum >= 10, "getMetrics columns");
    }
    { std::vector<double> gm, gr; L.gradStats(gm, gr); CHECK(gm.size() == gr.size() && !gm.empty() && gr[0] >= std::fabs(gm[0]), "gradStats"); }
    L.processStats(base, true, (unsigned)L.nGradSteps() + 1);
    L.processStats(base, false, (unsigned)L.nGradSteps() + 1);
    FILE* sf = std::fopen((base + "_stats.txt").c_str(), "r"); CHECK(sf != nullptr, "stats file");
    if (sf) {
      char row[2048]; int nRows = 0; std::string first;
      while (std::fgets(row, sizeof(row), sf)) { if (!nRows) first = row; ++nRows; }
      std::fclose(sf);
      CHECK(nRows == 3 && first.rfind("ID #/T   |  avgR", 0) == 0, "stats file: header once, then one line per call");
    }
  }
  // ---- RACER (Gaussian advantage head): acting stores V and the advantage of the drawn action, training follows the oracle ----
  {
    MDPdescriptor M2; M2.dimStateObserved = 9; M2.dimAction = 3; M2.bActionSpaceBounded = {true, false, true};
    HyperParameters H2; H2.learner = "RACER"; H2.nnLayerSizes = {64, 64}; H2.nnFunc = "Tanh"; H2.batchSize = 32; H2.maxTotObsNum = 5000;
    H2.clipImpWeight = 4; H2.epsAnneal = 0; H2.explNoise = 0.5; H2.outWeightsPrefac = 0.1; H2.randSeed = 77;
    VRACER R(M2, H2, 0);
    hl_config c2{}; c2.struct_size = sizeof(c2); c2.dimS = 9; c2.dimA = 3; c2.bounded[0] = 1; c2.bounded[2] = 1;
    c2.n_hidden = 2; c2.hidden[0] = c2.hidden[1] = 64; c2.nnFunc = HL_FUNC_TANH; c2.adv_kind = HL_ADV_GAUSSIAN;
    c2.batchSize = 32; c2.maxTotObsNum = 5000; c2.gamma = H2.gamma; c2.lambda = H2.lambda; c2.clipImpWeight = 4; c2.penalTol = H2.penalTol;
    c2.epsAnneal = 0; c2.learnrate = H2.learnrate; c2.nnLambda = H2.nnLambda; c2.explNoise = 0.5; c2.outWeightsPrefac = 0.1;
    c2.randSeed = 77; c2.n_ranks = 1; c2.ref_threads = 1;
    ol_learner* O2 = nullptr;
    CHECK(ol_create(&c2, &O2) == 0 && ol_init_weights(O2) == 0, "RACER oracle");
    // acting: 40 episodes of 25 steps through select(); the oracle gets the very same episodes (packed form)
    for (int e = 0; e < 40; ++e) {
      Agent agent(0, 1000 + e);
      for (int t = 0; t <= 25; ++t) {
        agent.agentStatus = t == 0 ? INIT : (t == 25 ? (e % 3 ? LAST : TERM) : CONT);
        agent.state.resize(9); for (int i = 0; i < 9; ++i) agent.state[i] = 0.3f * (float)std::sin(0.37 * t + i + e);
        agent.reward = 0.1 * std::cos(0.2 * t + e);
        R.select(agent);
      }
      const Fvec packed = R.packEpisode(0);
      CHECK(ol_append_packed_episode(O2, packed.data(), (int64_t)packed.size()) == 0, "oracle takes the packed episode");
    }
    {   // the stored advantage of a drawn action is not zero for this head, and equals the oracle's network view
      const Fvec packed = R.packEpisode(0);
      const size_t N = 26, tup = 9 + 1 + 3 + 6;
      double amax = 0; for (size_t t = 0; t + 1 < N; ++t) amax = std::max(amax, (double)std::fabs(packed[N * tup + 2 * N + t]));
      CHECK(amax > 0, "stored advantages are all zero");
    }
    R.initializeLearner(); CHECK(ol_initialize(O2) == 0, "RACER ol_initialize");
    std::vector<int64_t> f1(32), f2(32);
    for (int k = 1; k <= 20; ++k) {
      R.trainStep(1); CHECK(ol_step(O2, 1, nullptr) == 0, "RACER ol_step");
      hl_readback(R.handle(), HL_TAP_FLAT, f1.data(), 32 * 8); ol_readback(O2, HL_TAP_FLAT, f2.data(), 32 * 8);
      CHECK(f1 == f2, "RACER step %d: sampled indices differ", k);
    }
    const int64_t n2 = hl_num_params(R.handle());
    std::vector<float> a(n2), b(n2), t1(n2), t2(n2);
    hl_get_params(R.handle(), a.data(), t1.data(), t2.data()); ol_get_params(O2, b.data(), t1.data(), t2.data());
    CHECK(relinf(a, b) < 1e-4, "RACER weights after 20 steps: rel err %.3g", relinf(a, b));
    hl_scalars s2; ol_get_scalars(O2, &s2);
    CHECK(std::fabs(R.beta() - s2.beta) <= 1e-9 * s2.beta, "RACER beta");
    ol_destroy(O2);
  }
  // ---- RACER on a discrete action space: acting draws labels from the SoftPlus-normalised policy, training follows the oracle ----
  {
    MDPdescriptor M3; M3.dimStateObserved = 7; M3.dimAction = 1; M3.discreteActionValues = {5};
    HyperParameters H3; H3.learner = "RACER"; H3.nnLayerSizes = {32, 32}; H3.nnFunc = "SoftSign"; H3.batchSize = 32; H3.maxTotObsNum = 5000;
    H3.clipImpWeight = 4; H3.epsAnneal = 0; H3.outWeightsPrefac = 0.1; H3.randSeed = 99;
    VRACER D(M3, H3, 0);
    hl_config c3{}; c3.struct_size = sizeof(c3); c3.dimS = 7; c3.dimA = 1; c3.n_options = 5;
    c3.n_hidden = 2; c3.hidden[0] = c3.hidden[1] = 32; c3.nnFunc = HL_FUNC_SOFTSIGN; c3.adv_kind = HL_ADV_DISCRETE;
    c3.batchSize = 32; c3.maxTotObsNum = 5000; c3.gamma = H3.gamma; c3.lambda = H3.lambda; c3.clipImpWeight = 4; c3.penalTol = H3.penalTol;
    c3.epsAnneal = 0; c3.learnrate = H3.learnrate; c3.nnLambda = H3.nnLambda; c3.explNoise = H3.explNoise; c3.outWeightsPrefac = 0.1;
    c3.randSeed = 99; c3.n_ranks = 1; c3.ref_threads = 1;
    ol_learner* O3 = nullptr;
    CHECK(ol_create(&c3, &O3) == 0 && ol_init_weights(O3) == 0, "discrete oracle");
    int counts[5] = {0, 0, 0, 0, 0};
    for (int e = 0; e < 40; ++e) {
      Agent agent(0, 2000 + e);
      for (int t = 0; t <= 20; ++t) {
        agent.agentStatus = t == 0 ? INIT : (t == 20 ? (e % 2 ? LAST : TERM) : CONT);
        agent.state.resize(7); for (int i = 0; i < 7; ++i) agent.state[i] = 0.4f * (float)std::cos(0.23 * t + i - e);
        agent.reward = 0.2 * std::sin(0.3 * t + e);
        D.select(agent);
        if (t < 20) {
          const int label = (int)std::floor(agent.action[0]);
          CHECK(label >= 0 && label < 5 && agent.policyVector.size() == 5, "discrete action message");
          if (label >= 0 && label < 5) counts[label]++;
          double tot = 0; for (double p : agent.policyVector) tot += p;
          CHECK(std::fabs(tot - 1) < 1e-12, "policy vector sums to one");
        }
      }
      const Fvec packed = D.packEpisode(0);
      CHECK(ol_append_packed_episode(O3, packed.data(), (int64_t)packed.size()) == 0, "oracle takes the packed discrete episode");
    }
    CHECK(counts[0] > 0 && counts[1] > 0 && counts[2] > 0 && counts[3] > 0 && counts[4] > 0, "every option is drawn by a fresh policy");
    D.initializeLearner(); CHECK(ol_initialize(O3) == 0, "discrete ol_initialize");
    std::vector<int64_t> f1(32), f2(32);
    for (int k = 1; k <= 20; ++k) {
      D.trainStep(1); CHECK(ol_step(O3, 1, nullptr) == 0, "discrete ol_step");
      hl_readback(D.handle(), HL_TAP_FLAT, f1.data(), 32 * 8); ol_readback(O3, HL_TAP_FLAT, f2.data(), 32 * 8);
      CHECK(f1 == f2, "discrete step %d: sampled indices differ", k);
      if (getenv("HOST_PARITY_VERBOSE") && getenv("HOST_PARITY_VERBOSE")[0] == '2') {
        const int64_t nn = hl_num_params(D.handle());
        std::vector<float> wa(nn), wb(nn), m1a(nn), m1b(nn), m2a(nn), m2b(nn);
        hl_get_params(D.handle(), wa.data(), m1a.data(), m2a.data()); ol_get_params(O3, wb.data(), m1b.data(), m2b.data());
        int64_t worst = 0; double wd = 0;
        for (int64_t i = 0; i < nn; ++i) { const double d = std::fabs((double)wa[i] - wb[i]); if (d > wd) { wd = d; worst = i; } }
        std::printf("  step %d: W rel %.3g M1 rel %.3g M2 rel %.3g  worst W idx %lld of %lld (%.8g vs %.8g)\n", k, relinf(wa, wb), relinf(m1a, m1b),
                    relinf(m2a, m2b), (long long)worst, (long long)nn, wa[worst], wb[worst]);
      }
    }
    const int64_t n3 = hl_num_params(D.handle());
    std::vector<float> a(n3), b(n3), t1(n3), t2(n3);
    hl_get_params(D.handle(), a.data(), t1.data(), t2.data()); ol_get_params(O3, b.data(), t1.data(), t2.data());
    if (getenv("HOST_PARITY_VERBOSE")) {
      double sa = 0, sb = 0; for (int64_t i = 0; i < n3; ++i) { sa += (double)a[i] * (1 + i % 7); sb += (double)b[i] * (1 + i % 7); }
      std::printf("discrete weights after 20 steps: rel err %.3g  device sum %.12f oracle sum %.12f\n", relinf(a, b), sa, sb);
    }
    CHECK(relinf(a, b) < 1e-4, "discrete weights after 20 steps: rel err %.3g", relinf(a, b));
    ol_destroy(O3);
  }
  // ---- RACER on LSTM layers (RACER_RNN.json family): acting carries the episode's history, training follows the oracle ----
  {
    MDPdescriptor M4; M4.dimStateObserved = 4; M4.dimAction = 1; M4.bActionSpaceBounded = {true};
    HyperParameters H4; H4.learner = "RACER"; H4.nnType = "LSTM"; H4.nnBPTTseq = 6; H4.nnLayerSizes = {32, 32}; H4.nnFunc = "Tanh";
    H4.batchSize = 16; H4.maxTotObsNum = 5000; H4.clipImpWeight = 4; H4.epsAnneal = 0; H4.explNoise = 0.1; H4.outWeightsPrefac = 0.1;
    H4.gamma = 0.99; H4.nnLambda = 1e-6; H4.randSeed = 123;
    VRACER Rn(M4, H4, 0);
    hl_config c4{}; c4.struct_size = sizeof(c4); c4.dimS = 4; c4.dimA = 1; c4.bounded[0] = 1; c4.nn_type = HL_NN_LSTM; c4.nnBPTTseq = 6;
    c4.n_hidden = 2; c4.hidden[0] = c4.hidden[1] = 32; c4.nnFunc = HL_FUNC_TANH; c4.adv_kind = HL_ADV_GAUSSIAN;
    c4.batchSize = 16; c4.maxTotObsNum = 5000; c4.gamma = 0.99; c4.lambda = H4.lambda; c4.clipImpWeight = 4; c4.penalTol = H4.penalTol;
    c4.epsAnneal = 0; c4.learnrate = H4.learnrate; c4.nnLambda = 1e-6; c4.explNoise = 0.1; c4.outWeightsPrefac = 0.1;
    c4.randSeed = 123; c4.n_ranks = 1; c4.ref_threads = 1;
    ol_learner* O4 = nullptr;
    CHECK(ol_create(&c4, &O4) == 0 && ol_init_weights(O4) == 0, "LSTM oracle");
    for (int e = 0; e < 30; ++e) {
      Agent agent(0, 3000 + e);
      std::vector<float> hist;
      for (int t = 0; t <= 15; ++t) {
        agent.agentStatus = t == 0 ? INIT : (t == 15 ? (e % 2 ? LAST : TERM) : CONT);
        agent.state.resize(4); for (int i = 0; i < 4; ++i) agent.state[i] = 0.5f * (float)std::sin(0.41 * t + 1.3 * i + e);
        agent.reward = 0.1 * t;
        hist.insert(hist.end(), agent.state.begin(), agent.state.end());
        Rn.select(agent);
        if (t < 15 && e == 0) {      // the policy mean the agent acted on = the oracle's output for the same history window
          const int n = std::min(7, t + 1);
          std::vector<double> out(8);
          CHECK(ol_forward_sequence(O4, n, hist.data() + (size_t)(t + 1 - n) * 4, out.data()) == 0, "ol_forward_sequence");
          CHECK(std::fabs(agent.policyVector[0] - out[4]) <= 1e-5 * (1 + std::fabs(out[4])), "recurrent policy mean at t=%d: %.9g vs %.9g", t, agent.policyVector[0], out[4]);
        }
      }
      const Fvec packed = Rn.packEpisode(0);
      CHECK(ol_append_packed_episode(O4, packed.data(), (int64_t)packed.size()) == 0, "oracle takes the packed episode");
    }
    Rn.initializeLearner(); CHECK(ol_initialize(O4) == 0, "LSTM ol_initialize");
    std::vector<int64_t> f1(16), f2(16);
    for (int k = 1; k <= 20; ++k) {
      Rn.trainStep(1); CHECK(ol_step(O4, 1, nullptr) == 0, "LSTM ol_step");
      hl_readback(Rn.handle(), HL_TAP_FLAT, f1.data(), 16 * 8); ol_readback(O4, HL_TAP_FLAT, f2.data(), 16 * 8);
      CHECK(f1 == f2, "LSTM step %d: sampled indices differ", k);
    }
    const int64_t n4 = hl_num_params(Rn.handle());
    std::vector<float> a(n4), b(n4), t1(n4), t2(n4);
    hl_get_params(Rn.handle(), a.data(), t1.data(), t2.data()); ol_get_params(O4, b.data(), t1.data(), t2.data());
    CHECK(relinf(a, b) < 1e-4, "LSTM weights after 20 steps: rel err %.3g", relinf(a, b));
    ol_destroy(O4);
  }
  ol_destroy(O);
  std::printf(failures ? "host_parity: %d FAILURES\n" : "host_parity: OK\n", failures);
  return failures ? 1 : 0;
}
