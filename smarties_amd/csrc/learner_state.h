// smarties_amd/csrc/learner_state.h -- part of learner.cpp's ONE translation unit (included there, like step_exec.h): the learner's state (struct hl_learner), error / allocation / timing helpers, the network description built by hl_create
#pragma once

namespace {

inline long long roundUp(long long n, long long m) { return (n + m - 1) / m * m; }

struct EpMeta { int eid; long long off; int N; bool term; long long tag, ID; long long sampled = -1, agentID = 0; /* wire-format trailer, kept for byte-exact re-packing */ };

struct TimeRec { int name; hipEvent_t a, b; };

// one of the two minibatch workspaces + the indices of its GEMM problems in the device table
constexpr int PARAM_TAIL = 256;
struct StepBuf {
  DevBatch bt{}; float* X0 = nullptr;
  int segDxIdx = -1, segDxBlocks = 0;      // two recurrent layer types: the GEMM between the segments' backward passes
  std::vector<int> bigDw;                  // large batches: weight-gradient problems taken by big_dw_kernel (indices into the problem table)
  std::vector<int> fwdIdx, fwdBlocks, dxIdx, dxBlocks; int dwIdx = 0, dwAdamIdx = 0, dwCount = 0, dwBlocks = 0;
  int dwWideIdx = -1, dwWideAdamIdx = -1, dwWideBlocks = 0;      // recurrent nets: the same problems unsplit, for dw_wide_kernel (gemm16.hip)
  int splitMaxMN = 0;                      // > 0: some weight-gradient problems are split over the rows (largest M x N among them)
  DwTable dwTable{}, dwTableAdam{};        // the dW problems by value (kernel-argument table of dw_table_kernel)
};
struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int steps = 0; };

}  // namespace

struct hl_learner {
  hl_config cfg{};
  std::string err;
  int dev = 0;
  hipStream_t stream = nullptr;
  int dS = 0, dA = 0, B = 0, Bglobal = 0, nOut = 0, nDense = 0, nAdv = 0, nHidden = 0, Mmax = 0;
  // appended past observations / convolutional preprocessing (conv.hip): the network input is dIn = dS (1 + nApp) wide and
  // gathered by its own kernel; with convolutions hid[0] stands for the last convolutional layer (its X, Y, D, Dres are that
  // layer's), hid[1..] are the dense blocks behind it
  bool preproc = false; int dIn = 0, nApp = 0, nConv = 0;
  bool bigBatch = false;      // local batch above 1024 (sample.hip: big_sample_kernel)
  bool wideDw = true;         // recurrent nets: weight gradients over all (sample, step) rows as one launch without a split-row join (SMARTIES_HIP_GENERIC & 4: (tile, chunk) workgroups + splitk_reduce_kernel)
  bool convDxRide = true;       // ... and what of them needs no convolutional delta behind the unstrided layers' input-gradient launches (SMARTIES_HIP_GENERIC & 256: none rides)
  bool convDwDense = true;      // convolutional nets: those tiles inside the filter-gradient launch (SMARTIES_HIP_GENERIC & 256: a launch of their own)
  bool directDw = true; int directDwMinTiles = 128;      // weight-gradient launches of >= this many unsplit tiles take dw_wide_kernel's one-workgroup-per-tile form
  bool recFused = true;       // two LSTM layers of 32 cells: forward, head and backward of a sample as one launch (rec.hip: lstm32_step_wave_kernel; SMARTIES_HIP_GENERIC & 4: the three launches)
  bool panelHead = false;     // ... and headp.hip's 16-sample panels for the head (recurrent nets and local batches >= 2048)
  int bigMm = 0;              // ... with the kernels of bigmm.hip (bit 0: weight-stationary forward / dX panels, bit 1: weight gradients, bit 2: LDS-tiled forward / dX products, taken before the panels; SMARTIES_HIP_GENERIC & 128: none)
  std::vector<hl::GemmProblem> hostProbs;      // the problem table as the host built it (large batches: kernels taking a problem by value)
  // ... whose sampler draws the NEXT step's minibatch on a stream of its own while this step's launches run
  hipStream_t sideStream = nullptr; hipEvent_t evMain = nullptr, evSide = nullptr; bool sidePending = false;
  int extras = 0;      // state variables beyond the first convolution's image: a second input layer behind the conv stack (Approximator.cpp:249-259)
  bool convPrepStale = true;      // the filters' LDS layouts (ConvGeo::Wf, Wx) do not reflect W (conv.hip: conv_prep_kernel)
  ConvGeo cg[HL_MAX_CONV]{}; int convDwBlocks = 0;
  bool convRowsAtari = true;    // the first layer's row-block kernels with the RACER_atari geometry at compile time (SMARTIES_HIP_GENERIC & 16: any-geometry kernels)
  ConvTailPlan convTail{};      // convt.hip: sample-resident kernels for the layers behind the first (on = 0: per-layer launches)
  bool recTm = false; int* tmT = nullptr; int* tmSteps = nullptr; int* tmNext = nullptr;      // wide LSTM layers: time-step-major launches (rectm.hip)
  float* tmER[HL_MAX_HIDDEN] = {}; float* tmSD[HL_MAX_HIDDEN] = {}; float* tmFP[HL_MAX_HIDDEN] = {};
  unsigned* tmCtr = nullptr; int tmCtrN = 0; int tmCtrOff[HL_MAX_HIDDEN] = {}; float* tmET[HL_MAX_HIDDEN] = {};
  int tmMinCells = 64;      // layers wider than this: time-step-major
  bool recurrent = false; int recK = 0;    // LSTM hidden layers: rows per sample of the per-step buffers (nnBPTTseq + 1; one more for the time-step-major launches)
  int recWin = 0;                          // ... steps of a window: nnBPTTseq + 1
  // hl_config::encoder_rnn: the first recSplit recurrent layers are plain recurrent ("RNN") ones under MGU layers.  The window kernels
  // serve one layer type per launch: the stack runs as two segments, the lower one's outputs of EVERY window step are the upper one's
  // input rows (segY), the upper one's input errors the lower one's top errors (segDres)
  int nEncLayers = 0, recSplit = 0; float* segY = nullptr; float* segDres = nullptr; float* segScratch = nullptr; int ldSeg = 0;
  // convolutions in front of recurrent layers: the conv launches run over the B recK window rows (+ next states) of a minibatch
  // (rec.hip: window_rows_kernel); otherwise convB = B, convMmax = Mmax
  int convB = 0, convMmax = 0;
  long long* winSlot = nullptr; int* winT = nullptr; int* winNextSrc = nullptr; hl::DevScalars* scW = nullptr;
  RecLayer rec[HL_MAX_HIDDEN]{};
  int nOpt = 0, polDim = 0, nSig = 0;      // discrete head: options; entries of a stored policy (2 dA | nOpt); sigma ParamLayer size (dA | 0)
  long long maxObsLocal = 0, maxObsGlobal = 0, minObsLocal = 0;
  // parameter blob layout (Parameters::_computeNParams, Layers/Parameters.h:159-176)
  std::vector<long long> indW, nW, indB, nB;
  long long nParams = 0;
  float *W = nullptr, *M1 = nullptr, *M2 = nullptr, *G = nullptr;
  DevScalars* sc = nullptr;
  DevReplay rp{};
  StepBuf buf[2];                          // double-buffered minibatch workspace (see step_exec.h)
  int ldX0 = 0; int lastParity = 0;        // buffer used by the last executed step (taps)
  DevHidden hid[HL_MAX_HIDDEN];
  float* dOut = nullptr; int ldDo = 0;
  std::string episodeLog;                  // cumulative_rewards.dat of MemoryBuffer::pushBackEpisode (hl_set_episode_log)
  std::string logBase; long long gsCalls = 0;     // StatsTracker file (<logBase>_net_outGrad_stats.raw) and its nStep
  long long indWo = 0, indBo = 0, indBp = 0; int ldWo = 0;
  // gemm problem tables (device) + launch geometry
  GemmProblem* dProbs = nullptr;           // all GEMM problems of both buffers, contiguous
  float* splitPart = nullptr; size_t splitPartFloats = 0;   // partial tiles of the split weight-gradient problems
  float* widePart = nullptr; unsigned* wideCtr = nullptr; int wideTiles = 0;      // dw_wide_kernel (gemm16.hip): four partial tiles and an arrival counter per tile
  // replay bookkeeping (host)
  long long capSlots = 0; int capEps = 0;
  long long ringHead = 0;                  // next free slot
  std::deque<EpMeta> order;                // front = newest (position 0), back = oldest
  std::vector<int> freeEids; int nextEid = 0;
  std::vector<int> pendingRetrace;
  long long nTransitions = 0, nSeenSteps = 0, nSeenEps = 0, nGradSteps = 0;
  long long nGatheredB4Startup = INT64_MAX;
  bool tableDirty = true, countsDirty = true, initialized = false, inStep = false;
  // One lock per learner: every entry point takes it, so finished episodes (hl_append_episode) and rollout inference
  // (hl_forward) may come from env-service threads while the training thread steps (the reference's dataset_mutex,
  // ReplayMemory/MemoryBuffer.h:55; callers Core/Master.cpp:66-86).  Entry points only enqueue device work, so the lock
  // is held for microseconds except where a call has to wait for the device by its nature (read-backs).
  mutable std::recursive_mutex mu;
  // episode ingestion: two pinned host buffers filled in turn; a buffer is handed to ONE ingest kernel (which reads it
  // over the bus) when it is full or when the device state has to be current (flushPending)
  struct Staging { unsigned char* host = nullptr; size_t cap = 0, used = 0; int nEp = 0; hipEvent_t ev = nullptr; bool inFlight = false; };
  Staging stg[2]; int stgCur = 0; int tableCount = 0;      // tableCount: episodes in the table the device currently holds
  // ReplayStats::avgSquaredErr as the reference has it when episodes arrive (the pre-training error placeholder,
  // MemoryBuffer.cpp:486-487): the value of the last gradient step's statistics pass, taken BEFORE that step's removals;
  // 0 before the first step.  Computed on the device when needed (dStatsIns), at most once per step.
  double* dStatsIns = nullptr; bool statsFresh = false, anyStep = false;
  unsigned char* actPin = nullptr; unsigned actTag = 0; bool actFastOk = false;     // rollout inference of a few agents (hl_forward)
  // prioritised samplers (per.hip): probabilities / cumulative table of the stored transitions, rebuilt before every minibatch
  float *perProb = nullptr, *perKey = nullptr, *perKeyS = nullptr; double* perCp = nullptr; unsigned *perIdx = nullptr, *perIdxS = nullptr; void* perScan = nullptr; size_t perScanBytes = 0;
  void* perTemp = nullptr; size_t perTempBytes = 0; long long perCap = 0;
  // staging
  void* pinned = nullptr; size_t pinnedBytes = 0;
  long long* dFlatGiven = nullptr; int* dEidList = nullptr; int eidListCap = 0;
  float* dActS = nullptr; double* dActO = nullptr;     // staging of hl_forward: raw states in, outputs out [Mmax rows]
  bool stepChainOk = false;               // ... and the head and the input-gradient products with them (gemm16.hip: step_chain_kernel): the two-launch step for those networks
  bool chainOk = false; int chainHT = 0;  // the dense forward layers of a network off the fused path go out as one launch (gemm16.hip: fwd_chain_kernel)
  bool noConvReplay = false;            // (SMARTIES_HIP_GENERIC & 64) stack the minibatch rows (stack_gather_kernel) also when the first layer could read the replay
  mutable int minLen = 0; mutable long long minLenAtN = -1; mutable size_t minLenAtCount = 0;      // shortest stored episode (evictionDue, removal rules other than "oldest")
  bool noDeferBeta = false;             // (SMARTIES_HIP_GENERIC & 2) the whole bookkeeping stays in the dW launch
  float* dRedMax = nullptr; double* dRedErr = nullptr; int redCap = 0;
  double* dMomPartial = nullptr; double* dMoments = nullptr; int momBlocksCap = 0;
  double* dStatsOut = nullptr;
  // replayed graphs: one per entry of GRAPH_SIZES and starting minibatch buffer (step_exec.h)
  GraphSlot graphs[16][2]; bool graphsStale = false, useGraph = true;
  // SMARTIES_HIP_GENERIC (tests, comparisons): bits that make the learner take a GENERAL kernel / launch list where a specialised one
  // would serve -- every such route exists anyway for the shapes the specialised one does not cover; nothing else selects code paths
  //   1 no two-kernel fused step            2 no forward chain / activation kernel / deferred beta      4 recurrent: unfused launches, chunked dW
  //   8 conv: per-layer launches behind the first layer       16 conv: any-geometry kernels       32 conv: gather-form filter gradients
  //  64 conv: stacked rows, no row-block kernels              128 large batches: the common tile launches      256 weight-gradient tiles in launches of their own
  int generic = 0;
  bool plainGraph = false;      // the replayed steps of this net are stepEager's launches as graph nodes (the next minibatch's sampler in front): nets none of the rider forms serves
  // graphs of exactly n steps (hl_prepare_steps, or a call size seen three times in a row): the whole call is one launch
  // and its last node stamps a pinned host word, which hl_sync polls (tools/call_bench.hip)
  std::map<int, std::array<GraphSlot, 2>> exactGraphs;
  unsigned* notifyPin = nullptr; unsigned notifyIssued = 0; mutable bool tailNotify = false;
  int lastCallN = 0, sameCallN = 0;
  // the sampler of step k+1 rides along step k, also along the LAST step of a replayed graph: the next call finds its
  // minibatch ready in buffer preParity.  Whatever changes what a sampler sees (new episodes, evictions, explicit
  // indices, a generator read-out) first puts the generator back (dropPresample)
  bool preValid = false; int preParity = 0;
  int eagerChain = 3;                      // calls of up to this many plain steps are launched directly instead of as graphs
  long long nCollectives = 0;              // RCCL calls issued or captured so far (tests: every path speaks the same wire protocol)
  struct LayDesc { int type, nIn, size, ld; long long indW, indB; };   // 1 dense, 2 parametric residual, 3 ParamLayer, 4 LSTM, 5 MGU (ld = gates x cells), 6 convolution (nIn = filter floats, size = biases)
  std::vector<LayDesc> lay;       // trainable layers in network order (checkpoint packing, Network::save)
  bool exchGraph = true;     // replica exchanges may be captured into the replayed graphs (cleared if a capture fails)
  bool xcdSafe = false;      // fused kernel: panel exchange through agent-scope accesses (workgroup b was NOT found on XCD b % 8, or forced)
  bool fusedOk = false; unsigned* panelCtr = nullptr;   // fused forward/head/dX kernel (fused.hip) usable for this network
  bool foldOk = false, foldNow = false;    // ... and runs the exchange itself (round 6: dw_table_kernel's chunk workgroups); foldNow: for the launch being issued
  bool pushOk = false, pushGrad = false;   // replicas over peer windows: the weight-gradient launch pushes its tiles itself (PushArgs); pushGrad: for the launch being issued
  bool fusedWideOk = false;  // two equal hidden blocks with a wide state and / or a head beyond the fused kernel's: fusedw.hip takes the two-kernel step
  int dbgVariant = 0;
  // rccl
  ncclComm_t comm = nullptr;
  // one-kernel exchange through peer-mapped windows (xchg.hip)
  struct Xchg {
    bool on = false;
    unsigned char* win = nullptr; size_t winBytes = 0, slotsOffset = 0, slotBytes = 0;
    unsigned char** dPeers = nullptr; XchgCtl* ctl = nullptr;
    std::vector<void*> opened;               // windows opened through hipIpc (closed by hl_destroy)
    int maxChunks = XCHG_CHUNKS_NODE;             // chunk workgroups of a collective at most (fewer where replicas share a device: hl_xchg_connect)
  } xchg;
  // wait of the exchange kernel for a peer's message (SMARTIES_HIP_XCHG_TIMEOUT_MS): replicas are gated independently by their data
  // (blockGradientUpdates), so a peer may legitimately lag by seconds or minutes behind a slow simulator -- the reference's
  // MPI_Iallreduce simply waits.  Ten minutes (60 s killed a training run the reference would have carried on), then the
  // learner's sticky device error (the state stays as it was before that collective); tests and bench.py set their own shorter bound
  long long xchgTimeoutTicks = 60000000000LL;   // 600 s at 100 MHz (SMARTIES_HIP_XCHG_TIMEOUT_MS): how long a replica waits inside the exchange kernel for its peers
  // moments exchange state
  bool momentsPending = false, initPending = false;
  // timing
  bool timing = false; std::vector<std::string> tnames; std::vector<double> tsum; std::vector<long long> tcnt;
  std::vector<TimeRec> trecs;
};

namespace {

// Exchange windows (hl_xchg_export) are UNCACHED device memory, and uncached memory must never go back to the allocator: on this
// runtime (ROCm 7.2, gfx950) memory freed after a life as hipDeviceMallocUncached and handed out again by hipMalloc made kernels of
// LATER learners read stale values -- gradients off by whole tiles, a problem table with wild pointers (memory aperture violation);
// found in round 6 by the replica tests at the BASELINE shapes, which create and destroy dozens of learners in one process
// (tools/dbg_xchg3.py reproduces it: 7 of 8 iterations; never with the windows kept, nor with cached or fine-grained windows).
// A destroyed learner's window therefore waits here for the next learner that needs one of its size on its device.
struct WindowPool { std::mutex mu; std::multimap<std::pair<int, size_t>, unsigned char*> free; };
WindowPool& windowPool() { static WindowPool* p = new WindowPool; return *p; }      // (never destructed: the runtime may be gone by then)
unsigned char* windowPoolGet(int dev, size_t bytes) {
  WindowPool& wp = windowPool(); std::lock_guard<std::mutex> g(wp.mu);
  auto it = wp.free.find({dev, bytes});
  if (it == wp.free.end()) return nullptr;
  unsigned char* q = it->second; wp.free.erase(it); return q;
}
void windowPoolPut(int dev, size_t bytes, unsigned char* q) {
  WindowPool& wp = windowPool(); std::lock_guard<std::mutex> g(wp.mu);
  wp.free.insert({{dev, bytes}, q});
}

int fail(hl_learner* h, int code, const std::string& m) { if (h) h->err = m; return code; }
int hipFail(hl_learner* h, hipError_t e, const char* what) {
  return fail(h, HL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
// (every entry point may enqueue work behind the completion stamp of the last replayed call: hl_sync then has to ask the runtime)
#define HL_LOCK_RAW(h) std::lock_guard<std::recursive_mutex> hl_lock_guard__((h)->mu)
#define HL_LOCK(h) HL_LOCK_RAW(h); (h)->tailNotify = false
#define HIPCK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return hipFail(h, e__, #x); } while (0)
#define NCCLCK(x) do { ncclResult_t r__ = (x); if (r__ != ncclSuccess) return fail(h, HL_ERR_COMM, std::string(#x) + ": " + ncclGetErrorString(r__)); } while (0)

template <typename T> hipError_t devAlloc(T** p, size_t n) {
  hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
  if (e == hipSuccess) e = hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(T));
  // hipMemset of device memory returns before the fill has run (null stream), and the library's streams are non-blocking:
  // without this wait the zeros could land on top of what the first kernels on h->stream had already written
  // (seen as a 9 % flake of tests/cpp/host_parity: initializeLearner() followed at once by the first step)
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  return e;
}
template <typename T> hipError_t devGrow(T** p, size_t oldN, size_t newN, hipStream_t s) {
  T* q = nullptr;
  hipError_t e = devAlloc(&q, newN);
  if (e != hipSuccess) return e;
  if (*p) {
    e = hipStreamSynchronize(s); if (e != hipSuccess) return e;
    if (oldN) { e = hipMemcpy(q, *p, oldN * sizeof(T), hipMemcpyDeviceToDevice); if (e != hipSuccess) return e; }
    hipFree(*p);
  }
  *p = q; return hipSuccess;
}

int timerId(hl_learner* h, const char* name) {
  for (size_t i = 0; i < h->tnames.size(); ++i) if (h->tnames[i] == name) return (int)i;
  h->tnames.push_back(name); h->tsum.push_back(0); h->tcnt.push_back(0);
  return (int)h->tnames.size() - 1;
}
void timerFlush(hl_learner* h) {
  if (h->trecs.empty()) return;
  hipStreamSynchronize(h->stream);
  for (auto& r : h->trecs) {
    float ms = 0; hipEventElapsedTime(&ms, r.a, r.b);
    h->tsum[r.name] += ms; h->tcnt[r.name] += 1;
    hipEventDestroy(r.a); hipEventDestroy(r.b);
  }
  h->trecs.clear();
}
// run a launch, optionally bracketed by HIP events on the library's own stream
template <typename F> hipError_t timed(hl_learner* h, const char* name, hipStream_t st, F&& f) {
  if (!h->timing) return f();
  TimeRec r; r.name = timerId(h, name);
  hipEventCreate(&r.a); hipEventCreate(&r.b);
  hipEventRecord(r.a, st);
  hipError_t e = f();
  hipEventRecord(r.b, st);
  h->trecs.push_back(r);
  if (h->trecs.size() >= 4096) timerFlush(h);
  return e;
}

// ---- network description: same construction rules as the reference Builder --------------
// (Network/Builder.cpp:48-117 via Approximator::buildFromSettings and RACER::setupNet)
int buildNet(hl_learner* h) {
  const hl_config& c = h->cfg;
  h->indW.clear(); h->nW.clear(); h->indB.clear(); h->nB.clear();
  std::vector<long long> lw, lb;          // per layer requested sizes
  lw.push_back(0); lb.push_back(0);       // input layer
  int prev = c.dimS * (1 + c.nAppendedObs), nH = 0;
  struct Tmp { int nIn, size, hasRes; int denseLayer, resLayer; };
  std::vector<Tmp> hs;
  // Approximator::buildPreprocessing -> Builder::addConv2d (Approximator.cpp:231-271, Builder.cpp:172-215): SoftSign
  // convolutions right behind the input, no skip connections; filter KnC InC KnY KnX floats, one bias per output element
  std::vector<int> convLayer;
  for (int j = 0; j < c.n_conv; ++j) {
    const hl_conv2d& d = c.conv[j];
    convLayer.push_back((int)lw.size());
    lw.push_back((long long)d.outFeatures * d.inpFeatures * d.filtery * d.filterx); lb.push_back((long long)d.outFeatures * d.outY * d.outX);
    prev = d.outFeatures * d.outY * d.outX;
  }
  h->extras = 0;
  if (c.n_conv > 0) {      // InputLayer + JoinLayer (Builder.cpp:26-46): no parameters, two entries in the layer list; the join puts the extras first
    const int inAll = c.dimS * (1 + c.nAppendedObs), inImg = c.conv[0].inpFeatures * c.conv[0].inpY * c.conv[0].inpX;
    if (inAll > inImg) { h->extras = inAll - inImg; lw.push_back(0); lb.push_back(0); lw.push_back(0); lb.push_back(0); prev += h->extras; }
  }
  for (int j = 0; j < c.n_hidden; ++j) {
    if (c.hidden[j] <= 0) continue;
    Tmp t; t.nIn = prev; t.size = c.hidden[j]; t.denseLayer = (int)lw.size();
    const int ltype = (c.encoder_rnn && nH < h->nEncLayers) ? HL_NN_RNN : c.nn_type;      // ("RNN" encoder layers of a partially observable MDP, Approximator.cpp:264-270)
    const int gates = ltype == HL_NN_LSTM ? 4 : (ltype == HL_NN_MGU ? 2 : 0);     // Layer_LSTM.h:24-29, Layer_GRU.h:29-34
    if (gates) { lw.push_back((long long)gates * t.size * (t.nIn + t.size)); lb.push_back(gates * t.size); }
    else if (ltype == HL_NN_RNN) { lw.push_back(roundUp(t.size, 8) * (t.nIn + t.size)); lb.push_back(t.size); }   // [W_in; W_rec] (Layer_Base.h:24-28)
    else { lw.push_back(roundUp(t.size, 8) * t.nIn); lb.push_back(t.size); }
    t.hasRes = (t.denseLayer != 1);        // no skip connection after the first layer (Builder.cpp:89-95)
    t.resLayer = -1;
    if (t.hasRes) { t.resLayer = (int)lw.size(); lw.push_back(t.size); lb.push_back(t.size); }
    hs.push_back(t); prev = t.size; ++nH;
  }
  if (nH < 1) return HL_ERR_BAD_ARG;
  const int hOff = c.n_conv > 0 ? 1 : 0;      // hid[0] = the last convolutional layer
  if (nH + hOff > HL_MAX_HIDDEN) return HL_ERR_UNSUPPORTED;
  h->nHidden = nH + hOff;
  // VRACER: [V, mean]; RACER with the Gaussian advantage: [V, coef, L+, L-, mean] (RACER_common.cpp:172-186)
  // RACER discrete: [V, A x nOpt, logits x nOpt], no sigma layer (RACER_common.cpp:119-134)
  const bool discrete = c.adv_kind == HL_ADV_DISCRETE;
  h->nOpt = discrete ? c.n_options : 0; h->polDim = discrete ? c.n_options : 2 * c.dimA; h->nSig = discrete ? 0 : c.dimA;
  h->nAdv = c.adv_kind == HL_ADV_GAUSSIAN ? 1 + 2 * c.dimA : (discrete ? c.n_options : 0);
  h->nDense = 1 + h->nAdv + (discrete ? c.n_options : c.dimA); h->nOut = h->nDense + h->nSig;
  const int outLayer = (int)lw.size();
  lw.push_back(roundUp(h->nDense, 8) * prev); lb.push_back(h->nDense);
  const int paramLayer = h->nSig ? (int)lw.size() : -1;
  if (h->nSig) { lw.push_back(0); lb.push_back(h->nSig); }       // sigma ParamLayer (none behind a discrete policy)
  long long tot = 0;
  for (size_t l = 0; l < lw.size(); ++l) {
    h->indW.push_back(tot); h->nW.push_back(lw[l]); tot += roundUp(lw[l], 8);
    h->indB.push_back(tot); h->nB.push_back(lb[l]); tot += roundUp(lb[l], 8);
  }
  h->nParams = tot;
  h->nConv = c.n_conv;
  for (int j = 0; j < c.n_conv; ++j) {
    const hl_conv2d& d = c.conv[j]; ConvGeo& g = h->cg[j];
    g = ConvGeo{};
    g.InC = d.inpFeatures; g.InY = d.inpY; g.InX = d.inpX; g.KnC = d.outFeatures; g.KnY = d.filtery; g.KnX = d.filterx;
    g.S = d.stridex; g.OpY = d.outY; g.OpX = d.outX; g.K = g.InC * g.KnY * g.KnX; g.P = g.OpY * g.OpX;
    g.indW = h->indW[convLayer[j]]; g.indB = h->indB[convLayer[j]];
  }
  if (hOff) {
    const ConvGeo& g = h->cg[c.n_conv - 1];
    DevHidden& d = h->hid[0]; d = DevHidden{};
    d.nIn = g.K; d.size = g.KnC * g.P; d.ldW = 0; d.func = HL_FUNC_SOFTSIGN; d.hasRes = 0; d.resW = 0; d.lstm = 0;
    d.ldA = (int)roundUp(d.size + h->extras, 16);      // rows [extras | outputs of the last convolution]
  }
  for (int j = 0; j < nH; ++j) {
    DevHidden& d = h->hid[j + hOff];
    const int ltype = (c.encoder_rnn && j < h->nEncLayers) ? HL_NN_RNN : c.nn_type;
    d.nIn = hs[j].nIn; d.size = hs[j].size; d.lstm = ltype == HL_NN_LSTM ? 4 : (ltype == HL_NN_MGU ? 2 : (ltype == HL_NN_RNN ? 1 : 0));   // gates per cell (0: dense; 1: dense with a recurrent term)
    d.ldW = d.lstm >= 2 ? d.lstm * d.size : (int)roundUp(d.size, 8); d.func = c.nnFunc;
    d.indW = h->indW[hs[j].denseLayer]; d.indB = h->indB[hs[j].denseLayer];
    d.hasRes = hs[j].hasRes; d.resW = std::min(d.nIn, d.size);
    if (d.lstm >= 2 && d.hasRes && d.nIn < d.size) return HL_ERR_UNSUPPORTED;   // (the reference's residual would read LSTM cell states there, Layers.h:357)
    d.indWr = d.hasRes ? h->indW[hs[j].resLayer] : 0; d.indBr = d.hasRes ? h->indB[hs[j].resLayer] : 0;
    d.ldA = (int)roundUp(d.size, 16);
  }
  h->indWo = h->indW[outLayer]; h->indBo = h->indB[outLayer]; h->ldWo = (int)roundUp(h->nDense, 8);
  h->indBp = paramLayer >= 0 ? h->indB[paramLayer] : 0;
  h->lay.clear();
  for (int j = 0; j < c.n_conv; ++j) h->lay.push_back({6, (int)lw[convLayer[j]], (int)lb[convLayer[j]], 0, h->indW[convLayer[j]], h->indB[convLayer[j]]});
  for (int j = 0; j < nH; ++j) {
    const int ltype = (c.encoder_rnn && j < h->nEncLayers) ? HL_NN_RNN : c.nn_type;
    if (ltype == HL_NN_LSTM) h->lay.push_back({4, hs[j].nIn, hs[j].size, 4 * hs[j].size, h->indW[hs[j].denseLayer], h->indB[hs[j].denseLayer]});
    else if (ltype == HL_NN_MGU) h->lay.push_back({5, hs[j].nIn, hs[j].size, 2 * hs[j].size, h->indW[hs[j].denseLayer], h->indB[hs[j].denseLayer]});
    else if (ltype == HL_NN_RNN) h->lay.push_back({1, hs[j].nIn + hs[j].size, hs[j].size, (int)roundUp(hs[j].size, 8), h->indW[hs[j].denseLayer], h->indB[hs[j].denseLayer]});   // BaseLayer::save: input rows, then recurrent rows (Layer_Base.h:143-153)
    else h->lay.push_back({1, hs[j].nIn, hs[j].size, (int)roundUp(hs[j].size, 8), h->indW[hs[j].denseLayer], h->indB[hs[j].denseLayer]});
    if (hs[j].hasRes) h->lay.push_back({2, 0, hs[j].size, 0, h->indW[hs[j].resLayer], h->indB[hs[j].resLayer]});
  }
  h->lay.push_back({1, prev, h->nDense, h->ldWo, h->indWo, h->indBo});
  if (h->nSig) h->lay.push_back({3, 0, c.dimA, 0, 0, h->indBp});
  return HL_OK;
}

// std::mt19937 + libstdc++ uniform_real_distribution<float> for hl_init_weights (host, one-off)
struct HostMT {
  uint32_t x[624]; uint32_t p;
  void twist() {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    for (int k = 0; k < 624; ++k) {
      const uint32_t y = (x[k] & UP) | (x[(k + 1) % 624] & LO);
      x[k] = x[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    }
    p = 0;
  }
  uint32_t next() {
    if (p >= 624) twist();
    uint32_t z = x[p++];
    z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
    return z;
  }
};

int syncScalarsToHost(hl_learner* h, DevScalars* out) {
  HIPCK(hipMemcpyAsync(out, h->sc, sizeof(DevScalars), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  // sticky device-side error (a bounded in-kernel wait gave up: 77 = panel barrier of the fused
  // kernel, 78 = sampler -> gather hand-off): the results since then are not trustworthy
  if (out->errFlag != 0) {
    char msg[96]; snprintf(msg, sizeof(msg), "device-side failure code %d (in-kernel wait timed out)", out->errFlag);
    return fail(h, HL_ERR_HIP, msg);
  }
  return HL_OK;
}

int ensurePinned(hl_learner* h, size_t bytes) {
  if (bytes <= h->pinnedBytes) return HL_OK;
  HIPCK(hipStreamSynchronize(h->stream));
  if (h->pinned) hipHostFree(h->pinned);
  h->pinnedBytes = std::max(bytes, h->pinnedBytes * 2);
  HIPCK(hipHostMalloc(&h->pinned, h->pinnedBytes, hipHostMallocDefault));
  return HL_OK;
}

// Re-allocate the slot arrays with a larger capacity and re-pack the live episodes contiguously
// (oldest first), so that the FIFO ring is un-wrapped afterwards.  Rare: capacity is sized from
// maxTotObsNum at creation.
template <typename T> hipError_t repack(T** arr, size_t width, long long newCap,
                                        const std::deque<EpMeta>& order, hipStream_t s) {
  T* q = nullptr;
  hipError_t e = devAlloc(&q, (size_t)newCap * width);
  if (e != hipSuccess) return e;
  long long off = 0;
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    e = hipMemcpyAsync(q + (size_t)off * width, *arr + (size_t)it->off * width, (size_t)it->N * width * sizeof(T),
                       hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    off += it->N;
  }
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  if (*arr) hipFree(*arr);
  *arr = q;
  return hipSuccess;
}
int flushStaging(hl_learner* h);
// `compact`: same capacity, the live episodes re-packed contiguously (removal rules other than "oldest" leave holes
// inside the ring that only fall behind its tail when the oldest episode goes)
int growSlots(hl_learner* h, long long need, bool compact = false) {
  if (need <= h->capSlots && !compact) return HL_OK;
  if (h->rp.S) { int rc = flushStaging(h); if (rc) return rc; }     // staged episodes carry slot offsets of the present layout
  const long long newCap = compact ? h->capSlots : std::max(need, h->capSlots + h->capSlots / 2 + 4096);
  const int dS = h->dS, dA = h->dA; hipStream_t s = h->stream;
  HIPCK(repack(&h->rp.S, dS, newCap, h->order, s)); HIPCK(repack(&h->rp.A, dA, newCap, h->order, s));
  HIPCK(repack(&h->rp.MU, h->polDim, newCap, h->order, s)); HIPCK(repack(&h->rp.R, 1, newCap, h->order, s));
  HIPCK(repack(&h->rp.V, 1, newCap, h->order, s)); HIPCK(repack(&h->rp.ADV, 1, newCap, h->order, s));
  HIPCK(repack(&h->rp.RET, 1, newCap, h->order, s)); HIPCK(repack(&h->rp.DQ, 1, newCap, h->order, s));
  HIPCK(repack(&h->rp.IMPW, 1, newCap, h->order, s)); HIPCK(repack(&h->rp.DKL, 1, newCap, h->order, s));
  long long off = 0;
  for (auto it = h->order.rbegin(); it != h->order.rend(); ++it) {
    it->off = off; off += it->N;
    HIPCK(hipMemcpyAsync(h->rp.epOff + it->eid, &it->off, sizeof(long long), hipMemcpyHostToDevice, s));
  }
  HIPCK(hipStreamSynchronize(s));
  h->ringHead = off; h->capSlots = newCap; h->graphsStale = true;
  return HL_OK;
}
int growEpisodes(hl_learner* h, int need) {
  if (need <= h->capEps) return HL_OK;
  const int newCap = std::max(need, h->capEps * 2 + 1024);
  const size_t o = (size_t)h->capEps, n = (size_t)newCap;
  HIPCK(devGrow(&h->rp.epOff, o, n, h->stream)); HIPCK(devGrow(&h->rp.epN, o, n, h->stream));
  HIPCK(devGrow(&h->rp.epTerm, o, n, h->stream)); HIPCK(devGrow(&h->rp.epAgg, o * AGG_N, n * AGG_N, h->stream));
  HIPCK(devGrow(&h->rp.epTag, o, n, h->stream));
  HIPCK(devGrow(&h->rp.posRec, o + 1, n + 1, h->stream));
  HIPCK(devGrow(&h->rp.posEid, o, n, h->stream)); HIPCK(devGrow(&h->rp.posPrefix, o + 1, n + 1, h->stream));
  const size_t nFar = std::max<size_t>(n + 256, (size_t)FAR_REGS * 256);      // (the register walk reads FAR_REGS rows of 256 whatever the table holds)
  HIPCK(devGrow(&h->rp.farP, 0, nFar, h->stream)); HIPCK(devGrow(&h->rp.farN, 0, nFar, h->stream));
  h->capEps = newCap; h->graphsStale = true;
  return HL_OK;
}

// contiguous slot range for a new episode: FIFO ring over [0, capSlots)
int allocSlots(hl_learner* h, int N, long long* off) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (h->order.empty()) {
      if (N <= h->capSlots) { *off = 0; h->ringHead = N; return HL_OK; }
    } else {
      const long long head = h->ringHead, tail = h->order.back().off;   // oldest live episode starts at tail
      if (head > tail) {                       // live region [tail, head): free = [head, cap) and [0, tail)
        if (head + N <= h->capSlots) { *off = head; h->ringHead = head + N; return HL_OK; }
        if (N < tail) { *off = 0; h->ringHead = N; return HL_OK; }
      } else if (head + N < tail) {            // wrapped: free = [head, tail)
        *off = head; h->ringHead = head + N; return HL_OK;
      }
    }
    long long live = 0; for (const EpMeta& e : h->order) live += e.N;
    const bool holes = h->cfg.ERoldSeqFilter != HL_ER_OLDEST && live + N + 1 <= h->capSlots - h->capSlots / 16;
    int rc = holes ? growSlots(h, h->capSlots, true)                                          // squeeze the holes out
                   : growSlots(h, h->capSlots + std::max<long long>(N + 1, h->capSlots / 2));   // re-packs, un-wraps
    if (rc) return rc;
  }
  return fail(h, HL_ERR_STATE, "replay slot allocation failed");
}

int uploadTable(hl_learner* h) {
  const size_t nEp = h->order.size();
  int rc = growEpisodes(h, (int)nEp + 1); if (rc) return rc;
  const size_t bytes = (nEp + 1) * sizeof(PosRec) + (nEp + 1) * sizeof(long long) + nEp * sizeof(int) + 64;
  rc = ensurePinned(h, bytes); if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));   // the pinned buffer may still feed an earlier copy
  PosRec* rec = (PosRec*)h->pinned;
  long long* pre = (long long*)(rec + nEp + 1);
  int* pe = (int*)(pre + nEp + 1);
  long long acc = 0;
  for (size_t p = 0; p < nEp; ++p) {
    const EpMeta& e = h->order[p];
    pre[p] = acc; pe[p] = e.eid;
    rec[p].prefix = acc; rec[p].off = e.off; rec[p].tag = e.tag; rec[p].N = e.N;
    rec[p].eidTerm = e.eid | (e.term ? (int)0x80000000 : 0);
    acc += e.N - 1;
  }
  pre[nEp] = acc;
  rec[nEp].prefix = acc; rec[nEp].off = 0; rec[nEp].tag = -1; rec[nEp].N = 0; rec[nEp].eidTerm = 0;
  HIPCK(hipMemcpyAsync(h->rp.posRec, rec, (nEp + 1) * sizeof(PosRec), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->rp.posPrefix, pre, (nEp + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->rp.posEid, pe, nEp * sizeof(int), hipMemcpyHostToDevice, h->stream));
  h->tableDirty = false; h->tableCount = (int)nEp;
  return HL_OK;
}

int runSweep(hl_learner* h, const int* dEids, int count, int recompute, int skipRetrace = 0) {
  if (count <= 0) return HL_OK;
  const int nb = sweep_blocks(count);
  if (recompute && nb > h->redCap) {
    HIPCK(devGrow(&h->dRedMax, 0, (size_t)nb, h->stream));
    HIPCK(devGrow(&h->dRedErr, 0, (size_t)nb, h->stream));
    h->redCap = nb;
  }
  EpisodeSweepArgs a{}; a.sc = h->sc; a.rp = h->rp; a.eids = dEids; a.count = count;
  a.gamma = (float)h->cfg.gamma; a.lambda = (float)h->cfg.lambda; a.recompute = recompute; a.skipRetrace = skipRetrace;
  a.redMaxAbs = h->dRedMax; a.redErr = h->dRedErr; a.retKind = h->cfg.returnsEstimator;
  HIPCK(timed(h, recompute ? "episode_sweep_recompute" : "episode_sweep_retrace", h->stream,
              [&] { return launch_episode_sweep(a, nb, h->stream); }));
  // (a recompute sweep that also rewrites the estimates -- the 1000th-step pass over all episodes -- counts nsteps - 1 updates each)
  const bool rewrote = !skipRetrace && h->cfg.returnsEstimator != HL_RET_NONE;
  if (recompute) HIPCK(launch_far_build(h->rp, (int)h->order.size(), h->stream));
  if (recompute) HIPCK(launch_sweep_finish(h->sc, h->rp, h->dRedMax, h->dRedErr, rewrote ? (int)h->nTransitions : -1, nb, h->stream));
  return HL_OK;
}

int dropPresample(hl_learner* h);      // step_exec.h

// the statistics new episodes take their placeholder error from: over the table the device holds now (call before the
// table changes within a step)
int refreshInsertionStats(hl_learner* h) {
  if (h->statsFresh || !h->anyStep || h->tableCount <= 0) return HL_OK;
  HIPCK(launch_stats(h->sc, h->rp, h->tableCount, h->dStatsIns, h->stream));
  h->statsFresh = true;
  return HL_OK;
}
// hand the staged episodes to the ingest kernel (one launch for the whole batch) and switch to the other buffer
int flushStaging(hl_learner* h) {
  hl_learner::Staging& st = h->stg[h->stgCur];
  if (st.nEp == 0) return HL_OK;
  // placeholder error of the new episodes: the average squared error over the episodes the device table holds right now
  // (ReplayStats::avgSquaredErr as of the last statistics pass, MemoryBuffer.cpp:486-487)
  int rc = refreshInsertionStats(h); if (rc) return rc;
  IngestArgs ia{}; ia.rp = h->rp; ia.stage = st.host; ia.nEp = st.nEp; ia.dS = h->dS; ia.dA = h->dA; ia.polDim = h->polDim;
  ia.stats = h->dStatsIns; ia.nEpTable = h->anyStep ? h->tableCount : 0;
  HIPCK(timed(h, "ingest_kernel", h->stream, [&] { return launch_ingest(ia, h->stream); }));
  HIPCK(hipEventRecord(st.ev, h->stream));
  st.inFlight = true; st.nEp = 0; st.used = 0;
  h->stgCur ^= 1;
  return HL_OK;
}

// everything the host queued since the last step: table, counters, Retrace of new episodes
int flushPending(hl_learner* h) {
  if (h->tableDirty || h->countsDirty || !h->pendingRetrace.empty()) {   // a minibatch drawn ahead saw the old table
    int rc = dropPresample(h); if (rc) return rc;
  }
  { int rc = flushStaging(h); if (rc) return rc; }
  const bool tableChanged = h->tableDirty;
  if (h->tableDirty) { int rc = uploadTable(h); if (rc) return rc; }
  // the largest |TD error| over the stored episodes (MemoryProcessing.cpp:223, feeding ReplayStats::maxAbsError) is kept as a
  // running maximum by the bookkeeping pass: episodes that left take theirs along, new ones bring their placeholder error
  if (tableChanged && h->initialized && !h->order.empty()) HIPCK(launch_episode_max(h->sc, h->rp, (int)h->order.size(), h->stream));
  if (h->countsDirty) {
    HIPCK(launch_set_counts(h->sc, h->nTransitions, (long long)h->order.size(), h->nSeenEps, h->nSeenSteps, h->stream));
    h->countsDirty = false;
  }
  // the terms of the far-policy count are kept by table position (dev_common.h)
  if (tableChanged && !h->order.empty()) HIPCK(launch_far_build(h->rp, (int)h->order.size(), h->stream));
  if (!h->pendingRetrace.empty()) {
    const int n = (int)h->pendingRetrace.size();
    if (n > h->eidListCap) { HIPCK(devGrow(&h->dEidList, 0, (size_t)n * 2, h->stream)); h->eidListCap = n * 2; }
    HIPCK(hipMemcpyAsync(h->dEidList, h->pendingRetrace.data(), n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    int rc = runSweep(h, h->dEidList, n, 0); if (rc) return rc;
    h->pendingRetrace.clear();
  }
  return HL_OK;
}

}  // namespace
