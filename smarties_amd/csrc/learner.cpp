// smarties_amd/csrc/learner.cpp -- host side of libsmarties_hip.so: the hl_* C-ABI of
// include/smarties_hip.h.  Owns the device-resident replay buffer, the parameter blob, the
// per-step launch sequence (eager or as a replayed hipGraph) and the RCCL communicator.
// There is no CPU compute path in this library: every entry point that needs the GPU fails
// with HL_ERR_NO_DEVICE / HL_ERR_HIP if HIP is not usable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <atomic>
#include <unistd.h>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <sstream>
#include <iomanip>
#include <climits>
#include <limits>
#include <vector>

#include "kernels.h"

using namespace hl;

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
    int maxChunks = XCHG_CHUNKS;             // chunk workgroups of a collective at most (fewer where replicas share a device: hl_xchg_connect)
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

#include "step_exec.h"

// =================================================================================================
extern "C" {

int hl_version(void) { return 1; }
const char* hl_status_string(int s) {
  switch (s) { case HL_OK: return "HL_OK"; case HL_ERR_BAD_ARG: return "HL_ERR_BAD_ARG";
    case HL_ERR_NO_DEVICE: return "HL_ERR_NO_DEVICE"; case HL_ERR_HIP: return "HL_ERR_HIP";
    case HL_ERR_STATE: return "HL_ERR_STATE"; case HL_ERR_TOO_FEW_DATA: return "HL_ERR_TOO_FEW_DATA";
    case HL_ERR_COMM: return "HL_ERR_COMM"; case HL_ERR_IO: return "HL_ERR_IO";
    case HL_ERR_UNSUPPORTED: return "HL_ERR_UNSUPPORTED"; }
  return "HL_ERR_?";
}
const char* hl_last_error(const hl_learner* h) { return h ? h->err.c_str() : "null handle"; }

int hl_create(const hl_config* cfg, hl_learner** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(hl_config)) return HL_ERR_BAD_ARG;
  if (cfg->dimS <= 0 || cfg->dimA <= 0 || cfg->dimA > HL_MAX_DIMA || cfg->n_hidden < 1 ||
      cfg->n_hidden > HL_MAX_HIDDEN || cfg->batchSize <= 0 || cfg->n_ranks < 1) return HL_ERR_BAD_ARG;
  if (cfg->n_ranks > 256) return HL_ERR_UNSUPPORTED;   // the replica counters travel as 16-bit chunks in fp32 (tail_dev.h: encodeCounters)
  if (cfg->adv_kind != HL_ADV_ZERO && cfg->adv_kind != HL_ADV_GAUSSIAN && cfg->adv_kind != HL_ADV_DISCRETE) return HL_ERR_UNSUPPORTED;
  if (cfg->adv_kind == HL_ADV_DISCRETE && (cfg->dimA != 1 || cfg->n_options < 2 || cfg->n_options > 64)) return HL_ERR_BAD_ARG;   // head launch: one option per lane of the sample's wavefront, 1 + 2 x 64 staged deltas (above 32 options: not the panel / fused heads)
  if (cfg->nnFunc < HL_FUNC_LINEAR || cfg->nnFunc > HL_FUNC_EXP) return HL_ERR_UNSUPPORTED;   // the ten names of makeFunction (Functions.h:643-668)
  if (cfg->episode_order != HL_ORDER_STABLE) return HL_ERR_UNSUPPORTED;   // reference permutation: oracle only
  if (cfg->nn_type < HL_NN_FFNN || cfg->nn_type > HL_NN_RNN) return HL_ERR_UNSUPPORTED;
  if (cfg->returnsEstimator < HL_RET_RETRACE || cfg->returnsEstimator > HL_RET_NONE) return HL_ERR_BAD_ARG;
  if (cfg->nnOutputFunc < HL_FUNC_LINEAR || cfg->nnOutputFunc > HL_FUNC_EXP) return HL_ERR_UNSUPPORTED;
  if (cfg->n_encoder < 0 || cfg->n_encoder + cfg->n_hidden > HL_MAX_HIDDEN) return HL_ERR_BAD_ARG;
  if (cfg->nn_type != HL_NN_FFNN) {   // rec.hip: 256-thread workgroups looping over gates and cells (REC_GENC, REC_GENIN)
    if (cfg->n_conv <= 0 && (cfg->dimS > 256 || (long long)cfg->dimS * (1 + std::max(cfg->nAppendedObs, 0)) > 1024)) return HL_ERR_UNSUPPORTED;
    // (encoder layers are hidden layers of the same network, Learner_approximator.cpp:149-166: the same limits hold for them)
    // (LSTM / MGU layers of up to 1024 cells, all multiples of 16, without encoder layers or convolutions in front: the time-step-major
    //  launches -- rectm.hip -- serve their training windows and their acting windows; checked again where recTm is decided)
    const bool tmKind = (cfg->nn_type == HL_NN_LSTM || cfg->nn_type == HL_NN_MGU) && cfg->n_encoder == 0 && cfg->n_conv <= 0;
    for (int j = 0; j < cfg->n_hidden; ++j) if (cfg->hidden[j] > (tmKind && cfg->hidden[j] % 16 == 0 ? 1024 : 256)) return HL_ERR_UNSUPPORTED;
    for (int j = 0; j < cfg->n_encoder; ++j) if (cfg->encoder[j] > 256) return HL_ERR_UNSUPPORTED;
  }
  if (cfg->nAppendedObs < 0 || cfg->n_conv < 0 || cfg->n_conv > HL_MAX_CONV) return HL_ERR_BAD_ARG;
  if (cfg->encoder_rnn && cfg->nn_type != HL_NN_MGU) return HL_ERR_BAD_ARG;      // (the one mix the reference builds: Approximator.cpp:221-223, 264-270)
  if (cfg->ERoldSeqFilter < HL_ER_OLDEST || cfg->ERoldSeqFilter > HL_ER_MINERROR) return HL_ERR_BAD_ARG;
  if (cfg->dataSamplingAlgo < HL_SAMPLE_UNIFORM || cfg->dataSamplingAlgo > HL_SAMPLE_PERSEQ) return HL_ERR_BAD_ARG;
  for (int j = 0; j < cfg->n_conv; ++j) {   // each layer takes the previous one's image; the first one the whole stacked input
    const hl_conv2d& d = cfg->conv[j];
    const long long inSize = (long long)d.inpFeatures * d.inpY * d.inpX;
    const long long prev = j == 0 ? (long long)cfg->dimS * (1 + cfg->nAppendedObs)
                                  : (long long)cfg->conv[j - 1].outFeatures * cfg->conv[j - 1].outY * cfg->conv[j - 1].outX;
    if ((j == 0 ? inSize > prev : inSize != prev) || d.outFeatures < 1 || d.outY < 1 || d.outX < 1 || d.stridex < 1 || d.filterx < 1 || d.filtery < 1) return HL_ERR_BAD_ARG;
    if (d.outY != (d.inpY - d.filtery + 2 * d.paddiny) / d.stridey + 1 || d.outX != (d.inpX - d.filterx + 2 * d.paddinx) / d.stridex + 1) return HL_ERR_BAD_ARG;
    // conv.hip: zero padding, one power-of-two stride, <= 64 channels per layer, filters <= 32 wide
    if (d.paddinx || d.paddiny || d.stridex != d.stridey || (d.stridex & (d.stridex - 1)) || d.stridex > 8) return HL_ERR_UNSUPPORTED;
    if (d.outFeatures > 64 || d.inpFeatures > 64 || d.filterx > 31 || d.filtery > 63 || (long long)d.outFeatures * d.outY * d.outX >= (1 << 20)) return HL_ERR_UNSUPPORTED;
  }
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return HL_ERR_NO_DEVICE;
  hl_learner* h = new hl_learner();
  h->cfg = *cfg;
  if (cfg->n_encoder > 0) {     // createEncoder: the encoder layers are the first hidden layers of the one network (Learner_approximator.cpp:149-166)
    int n = 0;
    for (int j = 0; j < cfg->n_encoder; ++j) if (cfg->encoder[j] > 0) h->cfg.hidden[n++] = cfg->encoder[j];
    h->nEncLayers = n;
    for (int j = 0; j < cfg->n_hidden; ++j) h->cfg.hidden[n++] = cfg->hidden[j];
    h->cfg.n_hidden = n; h->cfg.n_encoder = 0;
  }
  h->dev = cfg->device_id >= 0 ? cfg->device_id : (cfg->rank % nDev);
  *out = h;   // so that the caller can read hl_last_error and must hl_destroy
  HIPCK(hipSetDevice(h->dev));
  {      // the kernels are written for gfx950's 160 KB of LDS per workgroup (rec.hip, conv.hip, fused.hip size their stages for it): refuse any other part here, not at the first launch
    int ldsMax = 0;
    HIPCK(hipDeviceGetAttribute(&ldsMax, hipDeviceAttributeMaxSharedMemoryPerBlock, h->dev));
    if (ldsMax < 160 * 1024) return fail(h, HL_ERR_NO_DEVICE, "device offers less than 160 KB of LDS per workgroup (the kernels are built for gfx950)");
  }
  HIPCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->dS = cfg->dimS; h->dA = cfg->dimA;
  const double nL = cfg->n_ranks;
  h->Bglobal = cfg->batchSize > 1 ? (int)(std::ceil(cfg->batchSize / nL) * nL) : cfg->batchSize;
  h->B = cfg->batchSize > 1 ? h->Bglobal / cfg->n_ranks : h->Bglobal;
  if (h->B > 16384) return fail(h, HL_ERR_UNSUPPORTED, "local batch > 16384");
  // 1024 < B <= 16384: one 1024-thread sampler workgroup (sample.hip: big_sample_kernel), states assembled by stack_gather_kernel,
  // one launch per layer and direction, weight gradients split over the rows, eager steps
  h->bigBatch = h->B > 1024;
  h->bigMm = h->bigBatch ? 7 : 0;
  h->panelHead = h->B >= 2048 || cfg->nn_type != HL_NN_FFNN;      // (measured: recurrent nets 68.5 -> 66.5 us per step at 2 x 32 cells; the 512-wide Atari head 152.5 -> 156.9: one wavefront set per sample there)
  if (const char* e = getenv("SMARTIES_HIP_GENERIC")) h->generic = atoi(e);
  if (h->generic & 4) { h->recFused = false; h->wideDw = false; }
  if (h->generic & 256) { h->convDxRide = false; h->convDwDense = false; }
  if (h->generic & 128) h->bigMm = 0;
  h->nApp = cfg->nAppendedObs; h->dIn = h->dS * (1 + h->nApp);
  // the states are gathered by stack_gather_kernel (conv.hip): appended observations, convolutions, large batches -- and states
  // wider than the 512 components the sampler's own gather stages (any width then; the fused steps are for narrower ones)
  h->preproc = h->nApp > 0 || cfg->n_conv > 0 || h->bigBatch || (cfg->nn_type == HL_NN_FFNN && h->dS > 512);
  if ((long long)h->dS * (1 + h->nApp) > (1 << 20)) return fail(h, HL_ERR_UNSUPPORTED, "more than 2^20 network inputs");
  for (int j = 0; j < h->cfg.n_hidden; ++j)      // (the merged list: encoder layers first)
    if (h->cfg.hidden[j] > 2048) return fail(h, HL_ERR_UNSUPPORTED, "hidden layer wider than 2048");
  h->maxObsGlobal = (long long)(std::ceil(cfg->maxTotObsNum / nL) * nL);
  h->maxObsLocal = h->maxObsGlobal / cfg->n_ranks;
  long long minObs = cfg->minTotObsNum <= 0 ? cfg->maxTotObsNum : cfg->minTotObsNum;
  minObs = std::min(minObs, (long long)cfg->maxTotObsNum);
  minObs = (long long)(std::ceil(minObs / nL) * nL);
  h->minObsLocal = minObs / cfg->n_ranks;
  int rc = buildNet(h); if (rc) return rc;
  const int B = h->B;
  h->Mmax = (int)roundUp(2 * B, 16);
  h->recurrent = cfg->nn_type != HL_NN_FFNN;
  if (h->recurrent) h->recK = h->recWin = (cfg->nnBPTTseq > 0 ? cfg->nnBPTTseq : 16) + 1;
  // LSTM / MGU layers wider than 64 cells: time-step-major launches (rectm.hip); the windows then carry the next state's step as a row
  // of their own (one row more per sample)
  if (h->recurrent && (cfg->nn_type == HL_NN_LSTM || cfg->nn_type == HL_NN_MGU) && !(h->generic & 4) && cfg->n_encoder == 0 && cfg->n_conv == 0) {
    bool wide = false, ok = true;
    // (measured again with a launch per diagonal, batch 128, 17 steps: LSTM 2 x 64 cells 314 us per-sample against 284 time-step-major, 2 x 48: 204 / 282;
    //  MGU 2 x 64: 238 / 488 -- the time-step-major chain has a floor of 37 (LSTM) / 74 (MGU) dependent launches)
    h->tmMinCells = cfg->nn_type == HL_NN_LSTM ? 48 : 64;
    for (int j = 0; j < h->cfg.n_hidden; ++j) { wide = wide || h->cfg.hidden[j] > h->tmMinCells; ok = ok && h->cfg.hidden[j] % 16 == 0; }
    // (the crossover, measured at batch 128 and 17 steps: 2 x 64 cells 325 us with the per-sample kernels against 452 time-step-major, 2 x 96: 631 against 499)
    h->recTm = wide && ok;
    if (h->recTm) h->recK += 1;
    // their weight gradients (2304 rows x 513 x 1024 per layer at 2 x 256 cells): the large-batch launch's 64 x 64 tiles over row chunks
    // (bigmm.hip: big_dw_kernel + the split-row join: 74 + 14 us against 197 of dw_wide_kernel's 16 x 16 tiles, which read their operands
    // 4 x as often)
    if (h->recTm && !(h->generic & 128)) h->bigMm |= 2;
  }
  if (h->recurrent && !h->recTm)
    for (int j = 0; j < h->cfg.n_hidden; ++j) if (h->cfg.hidden[j] > 256) return fail(h, HL_ERR_UNSUPPORTED, "recurrent layer wider than 256 cells: every layer must be a multiple of 16 cells (time-step-major launches)");
  h->convB = B; h->convMmax = h->Mmax;
  if (h->recurrent && h->nConv > 0) {
    if (h->nHidden < 2) return fail(h, HL_ERR_UNSUPPORTED, "recurrent network type behind convolutions without a recurrent layer (nnLayerSizes is empty)");
    // (the input gradient of the first recurrent layer is a GEMM over its gate deltas with 16-byte row loads; its input rows are staged in LDS)
    if ((std::max(h->hid[1].lstm, 1) * h->hid[1].size) % 4 != 0) return fail(h, HL_ERR_UNSUPPORTED, "recurrent layer behind convolutions: gates x cells must be a multiple of 4");
    if (h->hid[1].nIn > 1024) return fail(h, HL_ERR_UNSUPPORTED, "recurrent layer behind convolutions: more than 1024 inputs");
    h->convB = B * h->recK; h->convMmax = (int)roundUp((long long)h->convB + B, 16);
    HIPCK(devAlloc(&h->winSlot, (size_t)h->convB)); HIPCK(devAlloc(&h->winT, (size_t)h->convB)); HIPCK(devAlloc(&h->winNextSrc, (size_t)B));
    HIPCK(devAlloc(&h->scW, 1));
    h->plainGraph = true;     // (no riders on these nets' launches: their replayed steps are the eager launch list, captured -- step_exec.h: captureSteps)
  }
  if (h->bigBatch) {
    HIPCK(hipStreamCreateWithFlags(&h->sideStream, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&h->evMain, hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&h->evSide, hipEventDisableTiming));
  }
  // (+ PARAM_TAIL unused floats: scratch "bias" rows of weight-gradient problems that have no bias, step_exec.h)
  HIPCK(devAlloc(&h->W, (size_t)h->nParams + PARAM_TAIL)); HIPCK(devAlloc(&h->M1, (size_t)h->nParams + PARAM_TAIL));
  HIPCK(devAlloc(&h->M2, (size_t)h->nParams + PARAM_TAIL)); HIPCK(devAlloc(&h->G, (size_t)h->nParams + PARAM_TAIL));
  HIPCK(devAlloc(&h->sc, 1));
  h->ldX0 = (int)roundUp(h->dIn, 16);
  for (int j = 0; j < h->nHidden; ++j) {
    DevHidden& d = h->hid[j];
    const bool convOut = j == 0 && h->nConv > 0;      // the last convolution's activations: one row per conv row
    const size_t n = (size_t)(convOut ? h->convMmax : h->Mmax) * d.ldA, nd = (size_t)(convOut ? h->convB : B) * d.ldA;
    HIPCK(devAlloc(&d.X, n)); HIPCK(devAlloc(&d.Y, n));
    if (d.hasRes) HIPCK(devAlloc(&d.Rr, n)); else d.Rr = nullptr;
    HIPCK(devAlloc(&d.D, nd)); HIPCK(devAlloc(&d.Dres, nd));
  }
  {   // convolutional layers: activations per layer (the last one's are hid[0]'s), chunking of the filter-gradient reduction
    int blk = 0;
    for (int l = 0; l < h->nConv; ++l) {
      ConvGeo& g = h->cg[l];
      g.ldIn = l == 0 ? h->ldX0 : h->cg[l - 1].ldOut;
      g.ldOut = (int)roundUp((long long)g.KnC * g.P, 16);
      // (the last layer writes behind the extra state variables of its rows; its X / Y / D are accessed element-wise only)
      if (l == h->nConv - 1) { g.ldOut = h->hid[0].ldA; g.X = h->hid[0].X + h->extras; g.Y = h->hid[0].Y + h->extras; g.D = h->hid[0].D + h->extras; }
      else { HIPCK(devAlloc(&g.X, (size_t)h->convMmax * g.ldOut)); HIPCK(devAlloc(&g.Y, (size_t)h->convMmax * g.ldOut)); HIPCK(devAlloc(&g.D, (size_t)h->convB * g.ldOut)); }
      const long long R = (long long)h->convB * g.P;             // rows of the filter-gradient reduction
      const int tiles = ((g.K + 15) / 16) * ((g.KnC + 15) / 16);
      constexpr long long dwWgs = 640;      // (tile, chunk) workgroups per layer of the gather form (RACER_atari step, round 4: 139.3 us at 1024, 137.0 at 512 - 768, 140.4 at 256, 144.9 at 2048)
      long long rowsPer = std::max<long long>(64, (R * tiles + dwWgs - 1) / dwWgs);   // several hundred workgroups per layer
      rowsPer = std::min<long long>(roundUp(rowsPer, 16), 2048);
      g.chunkRows = (int)rowsPer; g.nChunks = (int)((R + rowsPer - 1) / rowsPer);
      // layers with a large input image (the first one of the Atari stacks): row-block kernels, one partial per (sample, row block)
      { int win = 0; const int rb = (h->generic & 64) ? 0 : conv_row_block(g, &win);
        g.rbRows = 0; g.rbCount = 0; g.rbWin = 0;
        if (l == 0 && rb > 0 && conv_rows_ok(g)) { g.rbRows = rb; g.rbCount = (g.OpY + rb - 1) / rb; g.rbWin = win; g.nChunks = h->convB * g.rbCount; } }
      // layers behind the first: both operands staged in LDS, one workgroup per (group of rows, 16 channels)
      g.dwG = (l > 0 && !g.rbRows && !(h->generic & 32)) ? conv_dw_staged_group(g, h->convB) : 0;
      if (g.dwG && (((uintptr_t)g.D | (uintptr_t)h->cg[l - 1].Y) & 15)) g.dwG = 0;      // (16-byte copies: the last layer's rows may start behind extra state variables)
      if (g.dwG) { g.nChunks = (h->convB + g.dwG - 1) / g.dwG; g.chunkRows = g.dwG * g.P; }
      g.dwBlock0 = blk; blk += g.rbRows ? 0 : (g.dwG ? g.nChunks * (g.KnC / 16) : g.nChunks * tiles);
      HIPCK(devAlloc(&g.part, (size_t)g.nChunks * g.KnC * g.K));
      HIPCK(devAlloc(&g.Wf, (size_t)conv_prep_floats(g, 0))); HIPCK(devAlloc(&g.Wx, (size_t)conv_prep_floats(g, 1)));
      if ((long long)h->convMmax * g.P >= (1ll << 31) || (long long)h->convMmax * g.InY * g.InX >= (1ll << 31)) return fail(h, HL_ERR_UNSUPPORTED, "convolution: rows x positions >= 2^31");
    }
    h->convDwBlocks = blk;
    if (h->nConv > 1 && !(h->generic & 8)) conv_tail_plan(h->cg, h->nConv, &h->convTail);
    if (h->generic & 16) { h->convTail.atari = 0; h->convRowsAtari = false; }
  }
  h->actFastOk = !h->recurrent && h->nConv == 0 && !(h->generic & 2);
  for (int j = 0; j < h->nHidden; ++j) if (h->hid[j].size > ACT_MAXW || h->hid[j].nIn > ACT_MAXW) h->actFastOk = false;
  if (h->recurrent && cfg->encoder_rnn && h->nEncLayers > 0 && h->nEncLayers < h->nHidden - (h->nConv > 0 ? 1 : 0)) {
    const int j0 = h->nConv > 0 ? 1 : 0; const DevHidden& top = h->hid[j0 + h->nEncLayers - 1]; const DevHidden& up = h->hid[j0 + h->nEncLayers];
    if ((up.lstm * up.size) % 4 != 0) return fail(h, HL_ERR_UNSUPPORTED, "MGU layer behind RNN encoder layers: an even number of cells is needed");
    h->recSplit = h->nEncLayers; h->ldSeg = (int)roundUp(top.size, 16);
    const size_t R = (size_t)B * h->recK;
    HIPCK(devAlloc(&h->segY, (R + B) * h->ldSeg)); HIPCK(devAlloc(&h->segDres, R * h->ldSeg)); HIPCK(devAlloc(&h->segScratch, R * h->ldSeg));
    h->plainGraph = true;     // (as for recurrent layers behind convolutions: the eager launch list, captured)
  }
  if (h->recurrent) {
    const size_t R = (size_t)B * h->recK;
    for (int j = h->nConv > 0 ? 1 : 0; j < h->nHidden; ++j) {      // (hid[0] of a convolutional net is its last convolution)
      const DevHidden& d = h->hid[j]; RecLayer& L = h->rec[j];
      L.nIn = d.nIn; L.nC = d.size; L.hasRes = d.hasRes; L.resW = d.resW; L.indW = d.indW; L.indB = d.indB; L.indWr = d.indWr; L.indBr = d.indBr;
      const size_t g = (size_t)d.lstm;         // gates per cell
      L.ldA = (int)roundUp(d.nIn + d.size + 1, 16); L.ldR = (int)roundUp(d.size, 16); L.ldA2 = (int)roundUp(d.size + 1, 16);
      // (+16: the float4 loads of the weight-gradient kernel start at column offsets inside a row -- MGU recurrent blocks,
      //  step_exec.h -- and may run past the end of the last row; the values are masked, the reads must stay in bounds)
      HIPCK(devAlloc(&L.A, R * L.ldA + 16)); HIPCK(devAlloc(&L.X, R * g * d.size)); HIPCK(devAlloc(&L.Y, R * g * d.size));
      HIPCK(devAlloc(&L.D, R * g * d.size + 16));
      if (d.hasRes) HIPCK(devAlloc(&L.Rd, R * L.ldR));
      if (d.lstm == 2) HIPCK(devAlloc(&L.A2, R * L.ldA2 + 16));
      if (h->recTm) { HIPCK(devAlloc(&h->tmER[j], (size_t)B * d.size)); HIPCK(devAlloc(&h->tmSD[j], (size_t)B * d.size)); HIPCK(devAlloc(&h->tmFP[j], (size_t)B * d.size)); HIPCK(devAlloc(&h->tmET[j], (size_t)B * d.size)); }
    }
    if (h->recTm) {
      HIPCK(devAlloc(&h->tmT, (size_t)B)); HIPCK(devAlloc(&h->tmSteps, (size_t)B)); HIPCK(devAlloc(&h->tmNext, (size_t)B));
      int nCtr = 0;
      for (int j = 0; j < h->nHidden; ++j) { h->tmCtrOff[j] = nCtr; nCtr += (h->hid[j].size / 16) * ((B + 15) / 16); }      // (recTm: no convolutions in front)
      HIPCK(devAlloc(&h->tmCtr, (size_t)nCtr)); h->tmCtrN = nCtr;
    }
  }
  {   // fused forward + head + dX kernel: two equal hidden blocks of width H <= 256, small state / action spaces
    const bool off = (h->generic & 1) != 0;
    if (!off && h->nHidden == 2 && !h->recurrent && !h->preproc) {
      const DevHidden& d0 = h->hid[0]; const DevHidden& d1 = h->hid[1];
      h->fusedOk = cfg->nnOutputFunc == HL_FUNC_LINEAR && d0.size == d1.size && d1.size >= 16 && d1.size <= 256 && (d1.size & (d1.size - 1)) == 0 && h->dS <= 32 && h->nAdv == 0 && h->nDense <= 8 && h->ldWo == 8 &&
                   !d0.hasRes && d1.hasRes && d1.nIn == d0.size && d0.func == d1.func &&
                   fused_lds_bytes(h->dS, d1.size) <= 160 * 1024;
    }
    if (h->fusedOk) {
      // where do the workgroups of a launch of the fused kernel's shape run?  Its panel exchange assumes that the H / 16
      // workgroups of a panel (same blockIdx % 8) share an XCD's L2
      const int HT = h->hid[1].size / 16, panels = (h->Mmax + 15) / 16, pg = (panels + 7) / 8, nBlk = 8 + 8 * HT * pg;
      int* dX = nullptr; HIPCK(devAlloc(&dX, (size_t)nBlk));
      HIPCK(launch_xcc_probe(nBlk, fused_threads(), fused_lds_bytes(h->dS, h->hid[1].size), dX, h->stream));
      std::vector<int> xcc((size_t)nBlk);
      HIPCK(hipMemcpyAsync(xcc.data(), dX, xcc.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCK(hipStreamSynchronize(h->stream)); hipFree(dX);
      bool same = true;
      for (int b = 8; b < nBlk; ++b) same = same && xcc[(size_t)b] == xcc[(size_t)(8 + ((b - 8) & 7))];
      const char* f = getenv("SMARTIES_HIP_PANEL_SAFE");
      h->xcdSafe = !same || (f && f[0] == '1');
      const size_t nCtr = (size_t)roundUp((h->Mmax + 15) / 16, 8) * 32;
      HIPCK(devAlloc(&h->panelCtr, nCtr));
      HIPCK(hipMemset(h->panelCtr, 0, nCtr * sizeof(unsigned)));
      HIPCK(hipStreamSynchronize(nullptr));
    }
  }
  if (!h->fusedOk && (h->nHidden == 2 || h->nHidden == 3) && !h->recurrent && !h->preproc && !(h->generic & 1)) {
    // the wide variant of the fused kernel (fusedw.hip): same placement, same probe; also for THREE equal hidden blocks (settings/RACER_glider.json)
    const DevHidden& d0 = h->hid[0]; const DevHidden& d1 = h->hid[1];
    const int comps = h->nOpt ? h->nOpt : h->dA, nLH = h->nHidden;
    bool ok = d0.size == d1.size && !d0.hasRes && d1.hasRes && d1.nIn == d0.size && d0.func == d1.func &&
              fused_wide_ok(h->dS, d1.size, h->nDense, h->nOut, h->ldWo, cfg->adv_kind == HL_ADV_GAUSSIAN ? h->nAdv : 0, comps, nLH);
    if (nLH == 3) { const DevHidden& d2 = h->hid[2]; ok = ok && d2.size == d1.size && d2.hasRes && d2.nIn == d1.size && d2.func == d1.func && !(h->generic & 512); }
    if (ok) {
      const int HT = d1.size / 16, panels = (h->Mmax + 15) / 16, pg = (panels + 7) / 8, nBlk = 8 + 8 * HT * pg;
      int* dX = nullptr; HIPCK(devAlloc(&dX, (size_t)nBlk));
      HIPCK(launch_xcc_probe(nBlk, fused_wide_threads(), fused_wide_lds_bytes(h->dS, d1.size, h->nDense, h->nOut, h->ldWo, cfg->adv_kind == HL_ADV_GAUSSIAN ? h->nAdv : 0, nLH), dX, h->stream));
      std::vector<int> xcc((size_t)nBlk);
      HIPCK(hipMemcpyAsync(xcc.data(), dX, xcc.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCK(hipStreamSynchronize(h->stream)); hipFree(dX);
      for (int b = 8; b < nBlk; ++b) ok = ok && xcc[(size_t)b] == xcc[(size_t)(8 + ((b - 8) & 7))];
      const char* f = getenv("SMARTIES_HIP_PANEL_SAFE");
      ok = ok && !(f && f[0] == '1');
    }
    if (ok) {
      const size_t nCtr = (size_t)roundUp((h->Mmax + 15) / 16, 8) * 32;
      HIPCK(devAlloc(&h->panelCtr, nCtr));
      h->fusedWideOk = true;
    }
  }
  h->noDeferBeta = (h->generic & 2) != 0;
  h->noConvReplay = (h->generic & 64) != 0;
  // networks off the fused path with two or more dense layers (short reductions): one forward launch if the groups of its
  // panels run where the kernel assumes (same probe as above, with that kernel's geometry)
  if (!h->fusedOk && !h->fusedWideOk && !h->recurrent && !h->bigBatch) {
    const int j0 = h->nConv > 0 ? 1 : 0;
    bool ok = h->nHidden - j0 >= 2 && !(h->generic & 2);
    int HT = 0;
    for (int j = j0; j < h->nHidden; ++j) { ok = ok && !gemm_oneshot_ok(GEMM_F, h->hid[j].nIn); HT = std::max(HT, (h->hid[j].size + 15) / 16); }
    ok = ok && HT <= 64;      // a panel's group waits for all its workgroups: they must fit one XCD at once (32 CUs x 3 workgroups at this kernel's registers)
    if (ok) {
      const int nBlk = fwd_chain_blocks(h->Mmax, HT);
      int* dX = nullptr; HIPCK(devAlloc(&dX, (size_t)nBlk));
      HIPCK(launch_xcc_probe(nBlk, 256, fwd_chain_lds_bytes(), dX, h->stream));
      std::vector<int> xcc((size_t)nBlk);
      HIPCK(hipMemcpyAsync(xcc.data(), dX, xcc.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCK(hipStreamSynchronize(h->stream)); hipFree(dX);
      for (int b = 8; b < nBlk; ++b) ok = ok && xcc[(size_t)b] == xcc[(size_t)(8 + ((b - 8) & 7))];
      const char* f = getenv("SMARTIES_HIP_PANEL_SAFE");
      ok = ok && !(f && f[0] == '1');
    }
    if (ok) {
      const size_t nCtr = (size_t)roundUp((h->Mmax + 15) / 16, 8) * 32;
      HIPCK(devAlloc(&h->panelCtr, nCtr));
      h->chainOk = true; h->chainHT = HT;
      // the head and the input gradients in the same launch: no convolutions in front, no layer whose input-gradient product takes the
      // long-reduction kernel (256 < units <= 640), one replica's gradient or a pushed one alike (launchWeightGrad)
      bool sc = h->nConv == 0 && !(h->generic & 512) && HT <= 32;      // (<= 512 units: a panel's group fits its XCD at one workgroup per CU, the kernel's registers at their worst)
      for (int j = 1; j < h->nHidden; ++j) sc = sc && !gemm_oneshot_ok(GEMM_X, h->hid[j].size);
      h->stepChainOk = sc;
    }
  }
  h->ldDo = (int)roundUp(h->nDense, 16);
  HIPCK(devAlloc(&h->dOut, (size_t)B * h->ldDo));
  for (int pb = 0; pb < 2; ++pb) {
    DevBatch& bt = h->buf[pb].bt;
    HIPCK(devAlloc(&h->buf[pb].X0, (size_t)h->convMmax * h->ldX0));
    HIPCK(devAlloc(&bt.flat, B)); HIPCK(devAlloc(&bt.pos, B)); HIPCK(devAlloc(&bt.eid, B)); HIPCK(devAlloc(&bt.t, B));
    HIPCK(devAlloc(&bt.slot, B)); HIPCK(devAlloc(&bt.nextOf, B)); HIPCK(devAlloc(&bt.nextSrc, B));
    HIPCK(devAlloc(&bt.tag, B)); HIPCK(devAlloc(&bt.pEid, B)); HIPCK(devAlloc(&bt.pNextOf, B));
    HIPCK(devAlloc(&bt.sVals, (size_t)std::max(B, 256)));
    HIPCK(devAlloc(&bt.O, (size_t)2 * B * h->nOut)); HIPCK(devAlloc(&bt.G, (size_t)B * h->nOut));
    HIPCK(devAlloc(&bt.rho, B)); HIPCK(devAlloc(&bt.dkl, B)); HIPCK(devAlloc(&bt.dq, B)); HIPCK(devAlloc(&bt.far, B));
    HIPCK(devAlloc(&bt.newDQ, B)); HIPCK(devAlloc(&bt.newDKL, B)); HIPCK(devAlloc(&bt.newW, B)); HIPCK(devAlloc(&bt.newV, B)); HIPCK(devAlloc(&bt.newQ, B));
    HIPCK(devAlloc(&bt.oldDQ, B)); HIPCK(devAlloc(&bt.oldDKL, B)); HIPCK(devAlloc(&bt.oldW, B)); HIPCK(devAlloc(&bt.oldV, B));
    HIPCK(devAlloc(&bt.oldADV, B)); HIPCK(devAlloc(&bt.nextV, B)); HIPCK(devAlloc(&bt.oldNextV, B));
    HIPCK(devAlloc(&bt.oldNextADV, B)); HIPCK(devAlloc(&bt.gParam, (size_t)B * std::max(h->nSig, 1)));
    HIPCK(devAlloc(&bt.aggIn, (size_t)B * AGG_N));
  }
  HIPCK(devAlloc(&h->dFlatGiven, B));
  HIPCK(devAlloc(&h->dMoments, (size_t)2 * h->dS + 3)); HIPCK(devAlloc(&h->dStatsOut, 16)); HIPCK(devAlloc(&h->dStatsIns, 16));
  HIPCK(devAlloc(&h->rp.stMean, h->dS)); HIPCK(devAlloc(&h->rp.stScale, h->dS)); HIPCK(devAlloc(&h->rp.stStd, h->dS));
  HIPCK(devAlloc(&h->rp.farStart, 4 * 256));
  {   // ring slack: an eighth of the budget plus room for the episodes in flight (bounded in bytes for image-sized states)
    const long long slack = std::max<long long>(512, std::min<long long>(8192, (64ll << 20) / ((long long)h->dS * 4)));
    rc = growSlots(h, h->maxObsLocal + h->maxObsLocal / 8 + slack); if (rc) return rc;
  }
  rc = growEpisodes(h, 4096); if (rc) return rc;
  // initial scalars (MemoryBuffer.h:41-44; Optimizer.h:96; ExecutionInfo.cpp:387,391)
  DevScalars s0; std::memset(&s0, 0, sizeof(s0));
  s0.beta = cfg->clipImpWeight <= 0 ? 1 : 1e-4; s0.alpha = 0.5;
  s0.Cmax = 1 + cfg->clipImpWeight; s0.Cinv = 1 / cfg->clipImpWeight;
  s0.adam_bt1 = 0.9; s0.adam_bt2 = 0.999; s0.rewMean = 0; s0.rewScale = 1; s0.rewStd = 1;
  s0.cntRetUpd = cfg->returnsEstimator == HL_RET_NONE ? -1 : 0;
  { HostMT g; uint32_t sd = (uint32_t)(cfg->randSeed + (uint64_t)cfg->rank); g.x[0] = sd;
    for (uint32_t i = 1; i < 624; ++i) g.x[i] = 1812433253u * (g.x[i - 1] ^ (g.x[i - 1] >> 30)) + i;
    g.p = 624;
    // a reference run with T OpenMP threads seeds T - 1 further generators from this one (ExecutionInfo.cpp:392-393): they feed
    // only the other threads' Adam noise seeds, but the T - 1 draws shift the stream every sample is drawn from
    for (int t = 1; t < cfg->ref_threads; ++t) (void)g.next();
    std::memcpy(s0.rng, g.x, sizeof(g.x)); s0.rngPos = g.p; }
  HIPCK(hipMemcpy(h->sc, &s0, sizeof(s0), hipMemcpyHostToDevice));
  std::vector<float> ones(h->dS, 1.f);
  HIPCK(hipMemcpy(h->rp.stScale, ones.data(), h->dS * sizeof(float), hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(h->rp.stStd, ones.data(), h->dS * sizeof(float), hipMemcpyHostToDevice));
  rc = buildProblems(h); if (rc) return rc;
  if (const char* e = getenv("SMARTIES_HIP_NO_GRAPH")) { if (e[0] == '1') h->useGraph = false; }
  if (const char* e = getenv("SMARTIES_HIP_XCHG_TIMEOUT_MS")) h->xchgTimeoutTicks = std::max(1LL, atoll(e)) * 100000LL;
  return HL_OK;
}

int hl_destroy(hl_learner* h) {
  if (!h) return HL_OK;
  h->mu.lock();      // (released before the handle goes away; no other thread may still be using it)
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->sideStream) hipStreamSynchronize(h->sideStream);
  timerFlush(h);
  invalidateGraphs(h);
  if (h->comm) ncclCommDestroy(h->comm);
  if (h->actPin) hipHostFree(h->actPin);
  if (h->notifyPin) hipHostFree(h->notifyPin);
  for (void* q : h->xchg.opened) hipIpcCloseMemHandle(q);
  if (h->xchg.win) { windowPoolPut(h->dev, h->xchg.winBytes, h->xchg.win); h->xchg.win = nullptr; }      // (never back to the allocator: windowPoolGet)
  for (void* q : {(void*)h->xchg.dPeers, (void*)h->xchg.ctl}) if (q) hipFree(q);
  void* ptrs[] = {h->splitPart, h->widePart, h->wideCtr, h->W, h->M1, h->M2, h->G, h->sc, h->dOut, h->dProbs, h->dFlatGiven, h->dEidList,
    h->dRedMax, h->dRedErr, h->dMomPartial, h->dMoments, h->dStatsOut, h->dStatsIns,
    h->rp.S, h->rp.A, h->rp.MU, h->rp.R, h->rp.V, h->rp.ADV, h->rp.RET, h->rp.DQ, h->rp.IMPW, h->rp.DKL,
    h->rp.epOff, h->rp.epN, h->rp.epTerm, h->rp.epAgg, h->rp.posEid, h->rp.posPrefix, h->rp.stMean, h->rp.stScale,
    h->rp.stStd, h->rp.epTag, h->rp.posRec, h->rp.farP, h->rp.farN, h->rp.farStart, h->panelCtr, h->dActS, h->dActO};
  for (int pb = 0; pb < 2; ++pb) {
    DevBatch& bt = h->buf[pb].bt;
    void* bp[] = {h->buf[pb].X0, bt.sVals, bt.tag, bt.pEid, bt.pNextOf, bt.flat, bt.pos, bt.eid, bt.t, bt.slot, bt.nextOf, bt.nextSrc,
      bt.O, bt.G, bt.rho, bt.dkl, bt.dq, bt.far, bt.newDQ, bt.newDKL, bt.newW, bt.newV, bt.newQ, bt.oldDQ, bt.oldDKL, bt.oldW,
      bt.oldV, bt.oldADV, bt.nextV, bt.oldNextV, bt.oldNextADV, bt.gParam, bt.aggIn};
    for (void* q : bp) if (q) hipFree(q);
  }
  for (void* q : {(void*)h->segY, (void*)h->segDres, (void*)h->segScratch}) if (q) hipFree(q);
  for (void* q : {(void*)h->winSlot, (void*)h->winT, (void*)h->winNextSrc, (void*)h->scW}) if (q) hipFree(q);
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) for (float* q : {h->rec[j].A, h->rec[j].X, h->rec[j].Y, h->rec[j].D, h->rec[j].Rd, h->rec[j].A2}) if (q) hipFree(q);
  for (int j = 0; j < HL_MAX_HIDDEN; ++j) for (float* q : {h->tmER[j], h->tmSD[j], h->tmFP[j], h->tmET[j]}) if (q) hipFree(q);
  for (void* q : {(void*)h->tmT, (void*)h->tmSteps, (void*)h->tmNext, (void*)h->tmCtr}) if (q) hipFree(q);
  for (void* p : ptrs) if (p) hipFree(p);
  for (int l = 0; l < h->nConv; ++l) { ConvGeo& g = h->cg[l];
    if (l != h->nConv - 1) for (float* p : {g.X, g.Y, g.D}) if (p) hipFree(p);
    for (float* p : {g.part, g.Wf, g.Wx}) if (p) hipFree(p); }
  for (int j = 0; j < h->nHidden; ++j) { DevHidden& d = h->hid[j];
    for (float* p : {d.X, d.Y, d.Rr, d.D, d.Dres}) if (p) hipFree(p); }
  if (h->pinned) hipHostFree(h->pinned);
  for (auto& st : h->stg) { if (st.host) hipHostFree(st.host); if (st.ev) hipEventDestroy(st.ev); }
  for (void* q : {(void*)h->perProb, (void*)h->perKey, (void*)h->perKeyS, (void*)h->perCp, (void*)h->perIdx, (void*)h->perIdxS, h->perTemp, h->perScan}) if (q) hipFree(q);
  if (h->sideStream) { hipStreamSynchronize(h->sideStream); hipStreamDestroy(h->sideStream); }
  if (h->evMain) hipEventDestroy(h->evMain);
  if (h->evSide) hipEventDestroy(h->evSide);
  if (h->stream) hipStreamDestroy(h->stream);
  h->mu.unlock();
  delete h; return HL_OK;
}

int64_t hl_num_params(const hl_learner* h) { return h ? h->nParams : -1; }
int32_t hl_num_outputs(const hl_learner* h) { return h ? h->nOut : -1; }
int32_t hl_num_layers(const hl_learner* h) { return h ? (int32_t)h->indW.size() : -1; }
int hl_param_layout(const hl_learner* h, int64_t* indW, int64_t* nW, int64_t* indB, int64_t* nB) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  for (size_t l = 0; l < h->indW.size(); ++l) {
    if (indW) indW[l] = h->indW[l];
    if (nW) nW[l] = h->nW[l];
    if (indB) indB[l] = h->indB[l];
    if (nB) nB[l] = h->nB[l];
  }
  return HL_OK;
}

// Layer::initialize in build order (Builder.cpp:131-137; Layer_Base.h:115-141; Layers.h:395-400,548-553)
int hl_init_weights(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = dropPresample(h); if (rc) return rc;
  DevScalars s; rc = syncScalarsToHost(h, &s); if (rc) return rc;
  HostMT g; std::memcpy(g.x, s.rng, sizeof(g.x)); g.p = s.rngPos;
  auto uni = [&](float a, float b) {
    float r = (float)g.next() / 4294967296.0f;
    if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
    return std::fmaf(r, b - a, a);   // the reference build contracts this into an FMA (-march with FMA)
  };
  auto initFactor = [&](int f, int inps, int outs) -> double {
    switch (f) { case HL_FUNC_LINEAR: return std::sqrt(1. / inps); case HL_FUNC_TANH: return std::sqrt(6. / (inps + outs));
      case HL_FUNC_SOFTSIGN: return std::sqrt(6.0 / (inps + outs)); case HL_FUNC_RELU: return std::sqrt(2. / inps);
      case HL_FUNC_LRELU: return std::sqrt(1.0 / inps); case HL_FUNC_SIGM: case HL_FUNC_HARDSIGN: return std::sqrt(6. / (inps + outs));
      case HL_FUNC_SOFTPLUS: case HL_FUNC_EXPPLUS: case HL_FUNC_EXP: return std::sqrt(2. / inps); }
    return 1;
  };
  std::vector<float> W((size_t)h->nParams, 0.f);
  for (int l = 0; l < h->nConv; ++l) {     // Conv2DLayer::initialize (Layer_Conv2D.h:198-213): fan-in InC KnX KnY, fan-out KnC, biases zero
    const ConvGeo& g = h->cg[l];
    const float init = (float)initFactor(HL_FUNC_SOFTSIGN, g.K, g.KnC);
    for (long long w = 0; w < (long long)g.KnC * g.K; ++w) W[g.indW + w] = uni(-init, init);
  }
  for (int j = h->nConv > 0 ? 1 : 0; j < h->nHidden; ++j) {
    const DevHidden& d = h->hid[j];
    const float fac = 1; const float init = fac * initFactor(d.func, d.nIn, d.size);
    if (d.lstm == 1) {   // BaseLayer::initialize with bRecurrent (Layer_Base.h:115-141): input weights, then the recurrent ones, one distribution
      for (int i = 0; i < d.nIn + d.size; ++i) for (int o = 0; o < d.size; ++o) W[d.indW + o + (long long)d.ldW * i] = uni(-init, init);
    } else if (d.lstm) {   // Layer_LSTM.h:167-185 / Layer_GRU.h:232-246: forget gates start open, input / output gates closed; weights in memory order
      const int nC = d.size;
      if (d.lstm == 4) for (int o = 0; o < nC; ++o) { W[d.indB + o] = 0.f; W[d.indB + nC + o] = -1.f; W[d.indB + 2 * nC + o] = 1.f; W[d.indB + 3 * nC + o] = -1.f; }
      else for (int o = 0; o < nC; ++o) { W[d.indB + o] = 1.f; W[d.indB + nC + o] = 0.f; }
      for (long long w = 0; w < (long long)d.lstm * nC * (d.nIn + nC); ++w) W[d.indW + w] = uni(-init, init);
    } else
    for (int i = 0; i < d.nIn; ++i) for (int o = 0; o < d.size; ++o) W[d.indW + o + (long long)d.ldW * i] = uni(-init, init);
    if (d.hasRes) for (int o = 0; o < d.size; ++o) { W[d.indWr + o] = 1.f; W[d.indBr + o] = 0.f; }
  }
  { const DevHidden& q = h->hid[h->nHidden - 1];
    const double iFac = h->cfg.outWeightsPrefac; const float fac = (iFac > 0) ? iFac : 1;
    const int oF = h->cfg.nnOutputFunc;
    const float init = fac * initFactor(oF, q.size, h->nDense);
    // Builder::setLastLayersBias (continuous actions, RACER_common.cpp:94-105): initial outputs {0 | Gaussian_advantage::setInitial
    // (Gaus_advantage.h:31-34) | zero means}; the layer stores their pre-images under nnOutputFunc (Function::inverse,
    // Layer_Base.h:122-125).  Discrete heads leave the biases at zero.
    auto inverse = [&](float in) -> float {
      switch (oF) { case HL_FUNC_TANH: return std::log((1 + in) / (1 - in)) / 2; case HL_FUNC_SIGM: return -std::log(1 / in - 1);
        case HL_FUNC_HARDSIGN: return in / std::sqrt(1 - in * in); case HL_FUNC_SOFTSIGN: return in / (1 - std::fabs(in));
        case HL_FUNC_LRELU: return in >= 0 ? in : in / 0.1f; case HL_FUNC_EXPPLUS: return std::log(std::exp(std::min(8.f, std::max(-8.f, in))) - 1);
        case HL_FUNC_SOFTPLUS: return (in * in - 0.25f) / in; case HL_FUNC_EXP: return std::log(in); default: return in; }
    };
    if (h->cfg.adv_kind != HL_ADV_DISCRETE) {
      std::vector<float> iv((size_t)h->nDense, 0.f);
      if (h->cfg.adv_kind == HL_ADV_GAUSSIAN) { iv[1] = -1.f; for (int e = 2; e < 1 + h->nAdv; ++e) iv[e] = 1.f; }
      for (int o = 0; o < h->nDense; ++o) {
        const float pre = inverse(iv[o]);
        if (!std::isfinite(pre)) return fail(h, HL_ERR_UNSUPPORTED, "nnOutputFunc has no finite pre-image of an initial output value (the reference starts from inf / nan there)");
        W[h->indBo + o] = pre;
      }
    }
    for (int i = 0; i < q.size; ++i) for (int o = 0; o < h->nDense; ++o) W[h->indWo + o + (long long)h->ldWo * i] = uni(-init, init);
    double S = h->cfg.explNoise; if (S < FLT_EPSILON) S = FLT_EPSILON;
    for (int o = 0; o < h->nSig; ++o) W[h->indBp + o] = (float)((S * S - 0.25) / S);   // SoftPlus::_inv (Functions.h:564-568)
  }
  HIPCK(hipMemcpyAsync(h->W, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  h->convPrepStale = true;
  std::memcpy(s.rng, g.x, sizeof(g.x)); s.rngPos = g.p;
  HIPCK(hipMemcpyAsync(&h->sc->rng[0], s.rng, sizeof(s.rng), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(&h->sc->rngPos, &s.rngPos, sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}

int hl_set_params(hl_learner* h, const float* w, const float* m1, const float* m2) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const size_t n = (size_t)h->nParams * sizeof(float);
  if (w) { HIPCK(hipMemcpyAsync(h->W, w, n, hipMemcpyHostToDevice, h->stream)); h->convPrepStale = true; }
  if (m1) HIPCK(hipMemcpyAsync(h->M1, m1, n, hipMemcpyHostToDevice, h->stream));
  if (m2) HIPCK(hipMemcpyAsync(h->M2, m2, n, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_get_params(hl_learner* h, float* w, float* m1, float* m2) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const size_t n = (size_t)h->nParams * sizeof(float);
  if (w) HIPCK(hipMemcpyAsync(w, h->W, n, hipMemcpyDeviceToHost, h->stream));
  if (m1) HIPCK(hipMemcpyAsync(m1, h->M1, n, hipMemcpyDeviceToHost, h->stream));
  if (m2) HIPCK(hipMemcpyAsync(m2, h->M2, n, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_set_rng_state(hl_learner* h, const uint32_t st[625]) {
  if (!h || !st) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  { int rc = dropPresample(h); if (rc) return rc; }
  HIPCK(hipMemcpyAsync(&h->sc->rng[0], st, 624 * 4, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(&h->sc->rngPos, st + 624, 4, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_get_rng_state(hl_learner* h, uint32_t st[625]) {
  if (!h || !st) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = dropPresample(h); if (rc) return rc;      // the state as of after the last executed step
  DevScalars s; rc = syncScalarsToHost(h, &s); if (rc) return rc;
  std::memcpy(st, s.rng, 624 * 4); st[624] = s.rngPos;
  return HL_OK;
}

// MemoryBuffer::addEpisodeToTrainingSet + Episode::finalize + pushBackEpisode
// (MemoryBuffer.cpp:131-170,479-520; Episode.cpp:244-274).  The host bookkeeping (slot range, episode id, counters, order)
// is immediate; the data is copied into the pinned staging buffer -- the caller's arrays are free after return -- and
// reaches HBM with the next ingest launch (flushStaging); the Retrace estimate follows on the stream.  No device wait here.
int hl_append_episode(hl_learner* h, int32_t N, const float* states, const double* actions, const double* mu,
                      const double* rewards, const float* values, const float* advantages, int32_t terminated,
                      int64_t tag) {
  if (!h || !states || !actions || !mu || !rewards || !values) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (N < 2) return fail(h, HL_ERR_BAD_ARG, "Episode must at least have s0 and sT");
  const int dS = h->dS, dA = h->dA, pD = h->polDim;
  long long off = 0; int rc = allocSlots(h, N, &off); if (rc) return rc;
  int eid;
  if (!h->freeEids.empty()) { eid = h->freeEids.back(); h->freeEids.pop_back(); }
  else { eid = h->nextEid++; rc = growEpisodes(h, eid + 1); if (rc) return rc; }
  const size_t nf = (size_t)N;
  auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t bS = al16(nf * dS * 4), bA = nf * dA * 8, bMU = nf * pD * 8, bR = nf * 8, bV = nf * 4, bADV = nf * 4;
  const size_t need = al16(bS + bA + bMU + bR + bV + bADV);
  const size_t head = al16(sizeof(IngestDesc) * INGEST_MAX_EP);
  hl_learner::Staging* st = &h->stg[h->stgCur];
  if (st->nEp >= INGEST_MAX_EP || (st->nEp > 0 && st->used + need > st->cap)) { rc = flushStaging(h); if (rc) return rc; st = &h->stg[h->stgCur]; }
  if (st->inFlight) { HIPCK(hipEventSynchronize(st->ev)); st->inFlight = false; }      // the kernel that read this buffer is done
  if (head + need > st->cap) {       // (an episode larger than the buffer: grow it; it holds nothing at this point)
    if (st->host) hipHostFree(st->host);
    st->cap = std::max<size_t>(head + need, (size_t)32 << 20); st->host = nullptr;
    HIPCK(hipHostMalloc((void**)&st->host, st->cap, hipHostMallocDefault));
    if (!st->ev) HIPCK(hipEventCreateWithFlags(&st->ev, hipEventDisableTiming));
  }
  if (st->nEp == 0) st->used = head;
  unsigned char* p = st->host + st->used;
  std::memcpy(p, states, nf * dS * 4); p += bS;
  std::memcpy(p, actions, bA); p += bA;
  std::memcpy(p, mu, bMU); p += bMU;
  std::memcpy(p, rewards, bR); p += bR;
  std::memcpy(p, values, bV); p += bV;
  if (advantages) std::memcpy(p, advantages, bADV); else std::memset(p, 0, bADV);
  double totR = 0;
  for (int t = 1; t < N; ++t) totR += rewards[t];
  IngestDesc& d = reinterpret_cast<IngestDesc*>(st->host)[st->nEp];
  d.off = off; d.tag = tag; d.data = st->used; d.N = N; d.eid = eid; d.term = terminated ? 1 : 0; d.totR = (float)totR;
  st->used += need; st->nEp += 1;
  // counters: storeAction increments for t = 1..N-2, ID taken before the final increment (:110,:167,:484)
  h->nSeenSteps += N - 2;
  const long long locTrain = h->nGatheredB4Startup == INT64_MAX ? -1 : h->nSeenSteps - h->nGatheredB4Startup;
  EpMeta e{eid, off, N, terminated != 0, tag, std::max(locTrain, (long long)0)};
  if (!h->episodeLog.empty()) {      // MemoryBuffer.cpp:492-503: "%ld %ld %d %u %f" = nGradSteps, time stamp, agent, steps, total reward (Fval)
    float totRf = 0; for (int t = 1; t < N; ++t) totRf = (float)((double)totRf + rewards[t]);      // Fval totR += Real reward (:95-96)
    if (FILE* f = std::fopen(h->episodeLog.c_str(), "a")) { std::fprintf(f, "%ld %ld %d %u %f\n", (long)h->nGradSteps, (long)e.ID, 0, (unsigned)N, totRf); std::fclose(f); }
    else { h->err = "unable to open " + h->episodeLog + " (episode log switched off)"; h->episodeLog.clear(); }   // the episode is staged: it enters the training set regardless
  }
  h->nSeenSteps += 1; h->nSeenEps += 1;
  h->order.push_front(e); h->minLenAtN = -1;
  h->nTransitions += N - 1;
  h->pendingRetrace.push_back(eid);
  h->tableDirty = true; h->countsDirty = true;
  return HL_OK;
}

int hl_get_scaling(hl_learner* h, float* m, float* sc, float* r3) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (m) HIPCK(hipMemcpyAsync(m, h->rp.stMean, h->dS * 4, hipMemcpyDeviceToHost, h->stream));
  if (sc) HIPCK(hipMemcpyAsync(sc, h->rp.stScale, h->dS * 4, hipMemcpyDeviceToHost, h->stream));
  DevScalars s; int rc = syncScalarsToHost(h, &s); if (rc) return rc;
  if (r3) { r3[0] = s.rewMean; r3[1] = s.rewScale; r3[2] = s.rewStd; }
  return HL_OK;
}
int hl_set_scaling(hl_learner* h, const float* m, const float* sc, const float* r3) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (m) HIPCK(hipMemcpyAsync(h->rp.stMean, m, h->dS * 4, hipMemcpyHostToDevice, h->stream));
  if (sc) {
    std::vector<float> sd(h->dS); for (int k = 0; k < h->dS; ++k) sd[k] = 1 / sc[k];
    HIPCK(hipMemcpyAsync(h->rp.stScale, sc, h->dS * 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(hipMemcpyAsync(h->rp.stStd, sd.data(), h->dS * 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
  }
  if (r3) {
    HIPCK(hipMemcpyAsync(&h->sc->rewMean, &r3[0], 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(hipMemcpyAsync(&h->sc->rewScale, &r3[1], 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(hipMemcpyAsync(&h->sc->rewStd, &r3[2], 4, hipMemcpyHostToDevice, h->stream));
  }
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_get_episode_info(hl_learner* h, int64_t pos, int64_t* tag, int32_t* nsteps, int32_t* term) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (pos < 0 || pos >= (int64_t)h->order.size()) return HL_ERR_BAD_ARG;
  const EpMeta& e = h->order[(size_t)pos];
  if (tag) *tag = e.tag;
  if (nsteps) *nsteps = e.N;
  if (term) *term = e.term;
  return HL_OK;
}
int hl_get_episode_stats(hl_learner* h, int64_t pos, float* dst) {
  if (!h || !dst) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (pos < 0 || pos >= (int64_t)h->order.size()) return HL_ERR_BAD_ARG;
  int rc = flushPending(h); if (rc) return rc;
  static_assert(AGG_TOTR == 0 && AGG_MINQ == 8, "the nine aggregates lead the record in Episode.h's order");
  HIPCK(hipMemcpyAsync(dst, h->rp.epAgg + (size_t)h->order[(size_t)pos].eid * AGG_N, 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_get_episode_field(hl_learner* h, int64_t pos, int32_t field, float* dst, int32_t cap) {
  if (!h || !dst) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (pos < 0 || pos >= (int64_t)h->order.size()) return HL_ERR_BAD_ARG;
  const EpMeta& e = h->order[(size_t)pos];
  if (cap < e.N) return HL_ERR_BAD_ARG;
  int rc = flushPending(h); if (rc) return rc;
  const float* src = nullptr;
  switch (field) { case HL_EP_RETURN: src = h->rp.RET; break; case HL_EP_VALUE: src = h->rp.V; break;
    case HL_EP_ADVANTAGE: src = h->rp.ADV; break; case HL_EP_IMPW: src = h->rp.IMPW; break;
    case HL_EP_DKL: src = h->rp.DKL; break; case HL_EP_DELTAQ: src = h->rp.DQ; break; default: return HL_ERR_BAD_ARG; }
  HIPCK(hipMemcpyAsync(dst, src + e.off, (size_t)e.N * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}

// Learner::initializeLearner (Learners/Learner.cpp:47-72), split at its two accurate reductions (updateCounters(true),
// updateRewardsStats(true): DelayedReductor::get(true) waits for them, so with several learners every one starts from the GLOBAL
// counters and reward / state moments).  hl_initialize does the exchange itself over the communicator of hl_xchg_connect /
// hl_comm_init; in host-exchange mode the caller sums hl_counters_exchange / hl_moments_exchange between the two halves.
static int initializeBegin(hl_learner* h) {
  if (h->order.empty()) return fail(h, HL_ERR_TOO_FEW_DATA, "empty replay");
  int rc = flushPending(h); if (rc) return rc;
  rc = launchMoments(h); if (rc) return rc;                          // updateRewardsStats(bInit): the local sums
  h->momentsPending = true; h->initPending = true;
  return HL_OK;
}
static int initializeEnd(hl_learner* h) {
  if (!h->initPending) return fail(h, HL_ERR_STATE, "hl_initialize_end without hl_initialize_begin");
  int rc = launchPost(h, 0, POST_INIT, h->stream); if (rc) return rc;     // updateCounters(bInit)
  rc = launchMomentsApply(h, true, 1); if (rc) return rc;
  h->momentsPending = false; h->initPending = false;
  h->nGatheredB4Startup = h->minObsLocal;
  rc = runSweep(h, nullptr, (int)h->order.size(), 0); if (rc) return rc;   // rescaleAllReturnEstimator
  HIPCK(hipStreamSynchronize(h->stream));
  h->initialized = true;
  if (h->graphsStale) { invalidateGraphs(h); h->graphsStale = false; }
  return captureAllGraphs(h);      // one-off costs belong here, not in the first training step
}
int hl_initialize_begin(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  return initializeBegin(h);
}
int hl_initialize_end(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  return initializeEnd(h);
}
int hl_initialize(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = initializeBegin(h); if (rc) return rc;
  rc = allreduceCounters(h); if (rc) return rc;
  rc = allreduceMoments(h); if (rc) return rc;
  return initializeEnd(h);
}

static int preStepChecks(hl_learner* h) {
  if (!h->initialized) return fail(h, HL_ERR_STATE, "step before hl_initialize");
  if (h->inStep) return fail(h, HL_ERR_STATE, "hl_step_begin called twice");
  if (h->nTransitions < h->B) return fail(h, HL_ERR_TOO_FEW_DATA, "Parameter minTotObsNum is too low for given problem");
  { const int rc = ensureConvPrep(h); if (rc) return rc; }
  return flushPending(h);
}

// StatsTracker (Utils/StatsTracker.cpp:28-107): mean / RMS of the output gradients of the last minibatch
static int gradStatsOfLastBatch(hl_learner* h, double* mean, double* rms) {
  const int B = h->B, nOut = h->nOut;
  std::vector<double> G((size_t)B * nOut);
  HIPCK(hipStreamSynchronize(h->stream));
  HIPCK(hipMemcpy(G.data(), h->buf[h->lastParity].bt.G, G.size() * sizeof(double), hipMemcpyDeviceToHost));
  const long double cnt = std::max((long double)2.2e-16, (long double)B);
  for (int o = 0; o < nOut; ++o) {
    long double a = 0, q = 0;
    for (int b = 0; b < B; ++b) { const long double g = G[(size_t)b * nOut + o]; a += g; q += g * g; }
    mean[o] = (double)(a / cnt); rms[o] = std::sqrt((double)(q / cnt));
  }
  return HL_OK;
}
int hl_grad_stats(hl_learner* h, double* mean, double* rms) {
  if (!h || !mean || !rms) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (h->gsCalls == 0 && !h->inStep) return fail(h, HL_ERR_STATE, "no gradient step yet");
  return gradStatsOfLastBatch(h, mean, rms);
}
int hl_set_episode_log(hl_learner* h, const char* path) { if (!h) return HL_ERR_BAD_ARG; HL_LOCK(h); h->episodeLog = path ? path : ""; return HL_OK; }
int hl_set_log_base(hl_learner* h, const char* base) { if (!h) return HL_ERR_BAD_ARG; HL_LOCK(h); h->logBase = base ? base : ""; return HL_OK; }
static int appendGradStats(hl_learner* h) {      // StatsTracker::printToFile (StatsTracker.cpp:65-85)
  if (h->cfg.rank != 0) return HL_OK;
  std::vector<double> m((size_t)h->nOut), r((size_t)h->nOut);
  int rc = gradStatsOfLastBatch(h, m.data(), r.data()); if (rc) return rc;
  const std::string name = h->logBase + "_net_outGrad_stats.raw";
  FILE* f = std::fopen(name.c_str(), h->gsCalls ? "ab" : "wb");
  if (!f) return fail(h, HL_ERR_IO, "unable to open " + name);
  if (!h->gsCalls) { const float hd = h->nOut + .1; std::fwrite(&hd, sizeof(float), 1, f); }
  std::vector<float> v(2 * (size_t)h->nOut);
  for (int o = 0; o < h->nOut; ++o) { v[o] = (float)m[o]; v[o + h->nOut] = (float)r[o]; }
  std::fwrite(v.data(), sizeof(float), v.size(), f); std::fclose(f);
  return HL_OK;
}

int hl_step(hl_learner* h, int32_t n, const int64_t* flat) {
  if (!h || n < 0) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int s = 0;
  if (n == h->lastCallN) { if (++h->sameCallN == 3 && !flat && n > h->eagerChain && n < 1000) { int rc = prepareExact(h, n); if (rc) return rc; } }
  else { h->lastCallN = n; h->sameCallN = 1; }
  while (s < n) {
    int rc = preStepChecks(h); if (rc) return rc;
    h->statsFresh = false; h->anyStep = true;      // the statistics new episodes see are those of the step now running
    const long long k = h->nGradSteps + 1;
    const bool logStep = !h->logBase.empty() && (h->nGradSteps % 1000) == 0;   // StatsTracker::printToFile turn
    const bool plain = !flat && (k % 1000) != 0 && !logStep && !evictionDue(h) && !h->timing && h->useGraph &&
                       h->cfg.dataSamplingAlgo == HL_SAMPLE_UNIFORM &&      // (the prioritised samplers rebuild their table before every minibatch)
                       (!exchanging(h) || (h->exchGraph && wired(h)));
    if (plain) {
      if (h->graphsStale) { invalidateGraphs(h); h->graphsStale = false; }
      // plain steps available before the next 1000-step sweep and within this call
      long long avail = std::min<long long>(n - s, 999 - (h->nGradSteps % 1000));
      if (h->bigBatch) avail = std::min<long long>(avail, 998 - (h->nGradSteps % 1000));      // (the step in front of a 1000th one draws nothing ahead: stepEager)
      int done = 0;
      if (avail > 0) { rc = replaySteps(h, avail, &done, s == 0 && avail == n); if (rc) return rc; }
      if (done > 0) { h->nGradSteps += done; h->gsCalls += done; s += done; continue; }
    }
    const long long* dFlat = nullptr;
    if (flat) {
      HIPCK(hipMemcpyAsync(h->dFlatGiven, flat + (size_t)s * h->B, h->B * sizeof(long long), hipMemcpyHostToDevice, h->stream));
      HIPCK(hipStreamSynchronize(h->stream));
      dFlat = h->dFlatGiven;
    }
    rc = stepEager(h, dFlat); if (rc) return rc;
    if (logStep) { rc = appendGradStats(h); if (rc) return rc; }
    h->gsCalls += 1;
    h->nGradSteps += 1; s += 1;
  }
  return HL_OK;
}

// split form (host-side exchange of gradient / counters / moments, e.g. over the existing MPI path)
int hl_step_begin(hl_learner* h, const int64_t* flat) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = preStepChecks(h); if (rc) return rc;
  h->statsFresh = false; h->anyStep = true;
  rc = dropPresample(h); if (rc) return rc;
  const long long* dFlat = nullptr;
  if (flat) {
    HIPCK(hipMemcpyAsync(h->dFlatGiven, flat, h->B * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    dFlat = h->dFlatGiven;
  }
  rc = launchSample(h, 0, dFlat, true, h->stream); if (rc) return rc;
  rc = launchMlp(h, 0, false, h->stream); if (rc) return rc;
  h->lastParity = 0;
  rc = launchPost(h, 0, POST_AGG, h->stream); if (rc) return rc;
  h->momentsPending = false;
  if (((h->nGradSteps + 1) % 1000) == 0) {
    rc = launchPeriodicSweep(h); if (rc) return rc;
    rc = launchMoments(h); if (rc) return rc;
    h->momentsPending = true;
  }
  rc = applyRemoval(h); if (rc) return rc;
  rc = flushPending(h); if (rc) return rc;
  h->inStep = true;
  return HL_OK;
}
int hl_grad_exchange(hl_learner* h, float* g, int32_t write_back) {
  if (!h || !g) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const size_t n = (size_t)h->nParams * sizeof(float);
  if (write_back) HIPCK(hipMemcpyAsync(h->G, g, n, hipMemcpyHostToDevice, h->stream));
  else HIPCK(hipMemcpyAsync(g, h->G, n, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_counters_exchange(hl_learner* h, int64_t c[4], int32_t write_back) {
  if (!h || !c) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (write_back) HIPCK(hipMemcpyAsync(h->sc->cnt, c, 4 * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  else HIPCK(hipMemcpyAsync(c, h->sc->cnt, 4 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_moments_exchange(hl_learner* h, double* io, int32_t write_back) {
  if (!h || !io) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (!h->momentsPending) return fail(h, HL_ERR_STATE, "no reward/state moments pending this step");
  const size_t n = (size_t)(2 * h->dS + 3) * sizeof(double);
  if (write_back) HIPCK(hipMemcpyAsync(h->dMoments, io, n, hipMemcpyHostToDevice, h->stream));
  else HIPCK(hipMemcpyAsync(io, h->dMoments, n, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}
int hl_step_end(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (!h->inStep) return fail(h, HL_ERR_STATE, "hl_step_end without hl_step_begin");
  int rc;
  if (h->momentsPending) { rc = launchMomentsApply(h, false, 10); if (rc) return rc; h->momentsPending = false; }
  rc = launchAdam(h, 0); if (rc) return rc;
  // the stand-alone Adam pass rewrites the filters but not their LDS layouts (only conv_reduce_adam_kernel with its fused Adam
  // keeps those current): the next forward rebuilds them (preStepChecks -> ensureConvPrep)
  if (h->nConv > 0) h->convPrepStale = true;
  rc = launchPost(h, 0, POST_BETA, h->stream); if (rc) return rc;
  if (!h->logBase.empty() && (h->nGradSteps % 1000) == 0) { rc = appendGradStats(h); if (rc) return rc; }
  h->gsCalls += 1;
  h->nGradSteps += 1; h->inStep = false;
  return HL_OK;
}
// ---- episodes in the reference's wire format (Episode::packEpisode / unpackEpisode, Episode.cpp:24-130) ----
int64_t hl_packed_episode_size(const hl_learner* h, int32_t N) {
  if (!h || N < 0) return -1;
  HL_LOCK(h);
  return (int64_t)(h->dS + h->dA + h->polDim + 1 + 6) * N + 10;      // Episode::computeTotalEpisodeSize (Episode.h:211-219)
}
int hl_append_packed_episode(hl_learner* h, const float* data, int64_t n) {
  if (!h || !data) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const int dS = h->dS, dA = h->dA, pD = h->polDim, tup = dS + 1 + dA + pD;
  const int64_t N = (n - 10) / (tup + 6);
  if (N < 2 || hl_packed_episode_size(h, (int32_t)N) != n) return fail(h, HL_ERR_BAD_ARG, "packed episode has the wrong size");
  std::vector<float> S((size_t)N * dS), V(N), ADV(N);
  std::vector<double> A((size_t)N * dA), MU((size_t)N * pD), R(N);
  const float* buf = data;
  for (int64_t i = 0; i < N; ++i) {      // Episode::unpackEpisode: fp32 -> Fvec states, Real reward, Rvec action / policy
    std::copy(buf, buf + dS, S.begin() + i * dS); R[i] = buf[dS]; buf += dS + 1;
    for (int j = 0; j < dA; ++j) A[i * dA + j] = buf[j];
    buf += dA;
    for (int j = 0; j < pD; ++j) MU[i * pD + j] = buf[j];
    buf += pD;
  }
  buf += N;                                            // returnEstimator: recomputed on insertion
  std::copy(buf, buf + N, ADV.begin()); buf += N;      // actionAdvantage
  std::copy(buf, buf + N, V.begin()); buf += N;        // stateValue
  buf += 3 * N;                                        // deltaValue, offPolicImpW, KullbLeibDiv: reset on insertion
  const char* cp = reinterpret_cast<const char*>(buf);
  bool term; int64_t ID; std::memcpy(&term, cp, sizeof(bool)); std::memcpy(&ID, cp + sizeof(bool), sizeof(int64_t));
  return hl_append_episode(h, (int32_t)N, S.data(), A.data(), MU.data(), R.data(), V.data(), ADV.data(), term ? 1 : 0, ID);
}
int hl_pack_episode(hl_learner* h, int64_t pos, float* dst, int64_t cap) {
  if (!h || !dst || pos < 0 || pos >= (int64_t)h->order.size()) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const EpMeta e = h->order[(size_t)pos];
  const int dS = h->dS, dA = h->dA, pD = h->polDim; const int64_t N = e.N, total = hl_packed_episode_size(h, e.N);
  if (cap < total) return fail(h, HL_ERR_BAD_ARG, "hl_pack_episode: destination too small");
  int rc = flushPending(h); if (rc) return rc;
  std::vector<float> S((size_t)N * dS), F((size_t)6 * N);
  std::vector<double> A((size_t)N * dA), MU((size_t)N * pD), R(N);
  HIPCK(hipMemcpyAsync(S.data(), h->rp.S + (size_t)e.off * dS, S.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipMemcpyAsync(A.data(), h->rp.A + (size_t)e.off * dA, A.size() * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipMemcpyAsync(MU.data(), h->rp.MU + (size_t)e.off * pD, MU.size() * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipMemcpyAsync(R.data(), h->rp.R + e.off, R.size() * 8, hipMemcpyDeviceToHost, h->stream));
  const float* src[6] = {h->rp.RET, h->rp.ADV, h->rp.V, h->rp.DQ, h->rp.IMPW, h->rp.DKL};   // order of Episode.cpp:48-72
  for (int k = 0; k < 6; ++k) HIPCK(hipMemcpyAsync(F.data() + (size_t)k * N, src[k] + e.off, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  std::fill(dst, dst + total, 0.f);
  float* buf = dst;
  for (int64_t i = 0; i < N; ++i) {
    std::copy(S.begin() + i * dS, S.begin() + (i + 1) * dS, buf); buf[dS] = (float)R[i]; buf += dS + 1;
    for (int j = 0; j < dA; ++j) buf[j] = (float)A[i * dA + j];
    buf += dA;
    for (int j = 0; j < pD; ++j) buf[j] = (float)MU[i * pD + j];
    buf += pD;
  }
  std::copy(F.begin(), F.end(), buf); buf += 6 * N;
  char* cp = reinterpret_cast<char*>(buf);
  const bool term = e.term; const int64_t ID = e.tag, sampled = e.sampled, agentID = e.agentID;
  std::memcpy(cp, &term, sizeof(bool)); cp += sizeof(bool);
  std::memcpy(cp, &ID, 8); cp += 8; std::memcpy(cp, &sampled, 8); cp += 8; std::memcpy(cp, &agentID, 8);
  return HL_OK;
}

// ---- statistics line (Learner::logStats: MemoryBuffer::getMetrics + AdamOptimizer::getMetrics) ----
static void real2SS(std::ostringstream& B, const double V, const int W, const bool bPos) {   // SstreamUtilities.h:51-63
  B << " " << std::setw(W);
  if (std::fabs(V) >= 1e4) B << std::setprecision(std::max(W - 7 + bPos, 0));
  else if (std::fabs(V) >= 1e3) B << std::setprecision(std::max(W - 6 + bPos, 0));
  else if (std::fabs(V) >= 1e2) B << std::setprecision(std::max(W - 5 + bPos, 0));
  else if (std::fabs(V) >= 1e1) B << std::setprecision(std::max(W - 4 + bPos, 0));
  else B << std::setprecision(std::max(W - 3 + bPos, 0));
  B << std::fixed << V;
}
int hl_metrics(hl_learner* h, char* header, int32_t headerCap, char* line, int32_t lineCap) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  hl_stats st; int rc = hl_get_stats(h, &st); if (rc) return rc;
  DevScalars sc; rc = syncScalarsToHost(h, &sc); if (rc) return rc;
  const bool qStats = st.minQ < st.maxQ;
  if (line) {
    std::ostringstream buff;
    real2SS(buff, st.avgReturn, 9, 0); real2SS(buff, (double)sc.rewMean, 6, 0); real2SS(buff, (double)sc.rewStd, 6, 1);
    real2SS(buff, st.avgKLdivergence, 5, 1);
    if (qStats) {
      const double EPS = std::numeric_limits<float>::epsilon();
      real2SS(buff, std::sqrt(std::max(EPS, st.avgSquaredErr)), 6, 1); real2SS(buff, st.maxAbsError, 6, 1);
      // the "dRet" column (MemoryBuffer.cpp:534-544): root-mean-square change of the return estimates in the sweeps since
      // the last line; printing consumes the counters
      long long newCnt = -1;
      if (st.countReturnsEstimateUpdates > 0) {
        const double nRet = (double)std::max<int64_t>(1, st.countReturnsEstimateUpdates), eRet = std::max(EPS, st.sumReturnsEstimateErrors);
        real2SS(buff, std::sqrt(eRet / nRet), 6, 1);
        newCnt = 0;
      }
      st.countReturnsEstimateUpdates = newCnt;
      HIPCK(launch_set_ret_counters(h->sc, newCnt, h->stream));
      real2SS(buff, st.stdevQ, 6, 1); real2SS(buff, st.avgQ, 6, 0); real2SS(buff, st.minQ, 6, 0); real2SS(buff, st.maxQ, 6, 0);
    }
    buff << " " << std::setw(5) << (long)h->order.size();
    buff << " " << std::setw(7) << (long)h->nTransitions;
    buff << " " << std::setw(7) << (long)sc.seenUpd[0];       // nSeenEps() / nSeenSteps(): as of the last updateCounters
    buff << " " << std::setw(8) << (long)sc.seenUpd[1];
    buff << " " << std::setw(7) << (long)st.nFarPolicySteps;
    if (sc.Cmax > 1) real2SS(buff, sc.beta, 6, 1);
    // AdamOptimizer::getMetrics: L2 norm of the whole (padded) weight blob in long double
    std::vector<float> w((size_t)h->nParams);
    rc = hl_get_params(h, w.data(), nullptr, nullptr); if (rc) return rc;
    long double sum = 0; for (float x : w) sum += (long double)x * (long double)x;
    real2SS(buff, (double)std::sqrt(sum), 7, 1);
    const std::string sLine = buff.str();
    if ((int)sLine.size() + 1 > lineCap) return fail(h, HL_ERR_BAD_ARG, "hl_metrics: line buffer too small");
    std::memcpy(line, sLine.c_str(), sLine.size() + 1);
  }
  if (header) {
    std::ostringstream buff;
    buff << "|  avgR  | avgr | stdr | DKL ";
    if (qStats) buff << (st.countReturnsEstimateUpdates >= 0 ? "| RMSE |maxErr| dRet | stdQ | avgQ | minQ | maxQ " : "| RMSE |maxErr| stdQ | avgQ | minQ | maxQ ");
    buff << "| nEp |  nObs | totEp | totObs | nFarP ";
    if (sc.Cmax > 1) buff << "| beta ";
    buff << std::left << std::setfill(' ') << "| " << std::setw(6) << "net";
    const std::string sHead = buff.str();
    if ((int)sHead.size() + 1 > headerCap) return fail(h, HL_ERR_BAD_ARG, "hl_metrics: header buffer too small");
    std::memcpy(header, sHead.c_str(), sHead.size() + 1);
  }
  return HL_OK;
}


// the bounds and the text block of MemoryProcessing::histogramImportanceWeights (MemoryProcessing.cpp:353-389)
static void impwBounds(float bounds[82]) {
  const int nBins = 81;
  const double beg = std::log(1e-3), end = std::log(50.0);
  bounds[0] = 0;
  for (int i = 1; i < nBins; ++i) bounds[i] = (float)std::exp(beg + (end - beg) * (i - 1.0) / (nBins - 2.0));
  bounds[nBins] = std::numeric_limits<float>::max() - 1e2;
}
static std::string impwText(const float bounds[82], const int64_t counts[81], double dataSize) {
  std::ostringstream buff;
  buff << "_____________________________________________________________________";
  buff << "\nOFF-POLICY IMP WEIGHTS HISTOGRAMS\n";
  buff << "weight pi/mu (harmonic mean of histogram's bounds):\n";
  for (int b = 0; b < 81; ++b) { const float x = bounds[b], y = bounds[b + 1]; real2SS(buff, 2 * x * (y / (x + y)), 6, 1); }
  buff << "\nfraction of dataset:\n";
  for (int b = 0; b < 81; ++b) real2SS(buff, counts[b] / dataSize, 6, 1);
  buff << "\n";
  buff << "_____________________________________________________________________";
  return buff.str();
}
int hl_impweight_histogram(hl_learner* h, char* text, int32_t cap, int64_t counts[HL_IMPW_BINS]) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = flushPending(h); if (rc) return rc;
  HistArgs ha{}; ha.rp = h->rp; ha.nEpisodes = (int)h->order.size(); impwBounds(ha.bounds);
  unsigned long long* dCnt = nullptr;
  HIPCK(devAlloc(&dCnt, 81));
  ha.counts = dCnt;
  HIPCK(launch_impw_hist(ha, h->stream));
  unsigned long long hc[81];
  HIPCK(hipMemcpyAsync(hc, dCnt, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  hipFree(dCnt);
  int64_t c64[81]; for (int b = 0; b < 81; ++b) c64[b] = (int64_t)hc[b];
  if (counts) std::memcpy(counts, c64, sizeof(c64));
  if (text) {
    const std::string t = impwText(ha.bounds, c64, (double)h->nTransitions);
    if ((int)t.size() + 1 > cap) return fail(h, HL_ERR_BAD_ARG, "hl_impweight_histogram: text buffer too small");
    std::memcpy(text, t.c_str(), t.size() + 1);
  }
  return HL_OK;
}

// ---- replay memory + ReF-ER state (MemoryBuffer::save / restart, MemoryBuffer.cpp:172-324) ----
static bool copyFile(const std::string& from, const std::string& to) {
  FILE* a = fopen(from.c_str(), "rb"); if (!a) return false;
  FILE* b = fopen(to.c_str(), "wb"); if (!b) { fclose(a); return false; }
  char buf[1 << 16]; size_t n;
  while ((n = fread(buf, 1, sizeof(buf), a)) > 0) fwrite(buf, 1, n, b);
  fclose(a); fclose(b); return true;
}
int hl_save_memory(hl_learner* h, const char* base, int32_t rank) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = flushPending(h); if (rc) return rc;
  DevScalars sc; rc = syncScalarsToHost(h, &sc); if (rc) return rc;
  const int dS = h->dS;
  std::vector<float> mean(dS), scale(dS), stdv(dS);
  HIPCK(hipMemcpy(mean.data(), h->rp.stMean, dS * 4, hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(scale.data(), h->rp.stScale, dS * 4, hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(stdv.data(), h->rp.stStd, dS * 4, hipMemcpyDeviceToHost));
  const std::string B(base);
  {
    const std::string back = B + "_scaling_backup.raw";
    FILE* f = fopen(back.c_str(), "wb"); if (!f) return fail(h, HL_ERR_IO, "Unable to save into file " + back);
    std::vector<double> V(mean.begin(), mean.end()); fwrite(V.data(), 8, V.size(), f);
    V.assign(scale.begin(), scale.end()); fwrite(V.data(), 8, V.size(), f);
    V.assign(stdv.begin(), stdv.end()); fwrite(V.data(), 8, V.size(), f);
    const double r3[3] = {(double)sc.rewStd, (double)sc.rewScale, (double)sc.rewMean};
    fwrite(r3, 8, 3, f); fclose(f);
    copyFile(back, B + "_scaling.raw");
  }
  char rk[64]; snprintf(rk, sizeof(rk), "_rank_%03u_learner_", (unsigned)rank);
  const std::string fName = B + rk;
  {
    FILE* f = fopen((fName + "status_backup.raw").c_str(), "w"); if (!f) return fail(h, HL_ERR_IO, "Unable to save into file " + fName);
    fprintf(f, "nStoredEps: %lu\n", (unsigned long)h->order.size());
    fprintf(f, "nStoredObs: %lu\n", (unsigned long)h->nTransitions);
    fprintf(f, "nLocalSeenEps: %lu\n", (unsigned long)h->nSeenEps);
    fprintf(f, "nLocalSeenObs: %lu\n", (unsigned long)h->nSeenSteps);
    fprintf(f, "nInitialData: %ld\n", (long)h->nGatheredB4Startup);
    fprintf(f, "nGradSteps: %ld\n", (long)(h->nGradSteps + 1));           // the reference writes counters.nGradSteps + 1
    fprintf(f, "CmaxReFER: %le\n", sc.Cmax);
    fprintf(f, "beta: %le\n", sc.beta);
    fclose(f);
  }
  {
    FILE* f = fopen((fName + "data_backup.raw").c_str(), "wb"); if (!f) return fail(h, HL_ERR_IO, "Unable to save into file " + fName);
    std::vector<float> buf;
    for (long long p = (long long)h->order.size() - 1; p >= 0; --p) {      // oldest first: re-appending restores the order
      const unsigned long N = (unsigned long)h->order[(size_t)p].N;
      buf.resize((size_t)hl_packed_episode_size(h, (int32_t)N));
      rc = hl_pack_episode(h, p, buf.data(), (int64_t)buf.size()); if (rc) { fclose(f); return rc; }
      fwrite(&N, sizeof(unsigned long), 1, f); fwrite(buf.data(), 4, buf.size(), f);
    }
    fclose(f);
  }
  copyFile(fName + "status_backup.raw", fName + "status.raw");
  copyFile(fName + "data_backup.raw", fName + "data.raw");
  return HL_OK;
}
int hl_restart_memory(hl_learner* h, const char* base, int32_t rank) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (!h->order.empty()) return fail(h, HL_ERR_STATE, "hl_restart_memory needs an empty replay");
  const int dS = h->dS, dA = h->dA;
  const std::string B(base);
  {
    FILE* f = fopen((B + "_scaling.raw").c_str(), "rb");
    if (!f) return fail(h, HL_ERR_IO, "Parameters restart file " + B + "_scaling.raw not found.");
    std::vector<double> V((size_t)3 * dS + 3);
    const size_t got = fread(V.data(), 8, V.size(), f); fclose(f);
    if (got != V.size()) return fail(h, HL_ERR_IO, "Mismatch in restarted file " + B + "_scaling.raw");
    std::vector<float> m(dS), s(dS), d(dS);
    for (int i = 0; i < dS; ++i) { m[i] = (float)V[i]; s[i] = (float)V[dS + i]; d[i] = (float)V[2 * dS + i]; }
    HIPCK(hipMemcpy(h->rp.stMean, m.data(), dS * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->rp.stScale, s.data(), dS * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->rp.stStd, d.data(), dS * 4, hipMemcpyHostToDevice));
    const float r3[3] = {(float)V[3 * dS + 2], (float)V[3 * dS + 1], (float)V[3 * dS]};   // mean, scale, std
    HIPCK(hipMemcpy(&h->sc->rewMean, &r3[0], 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&h->sc->rewScale, &r3[1], 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&h->sc->rewStd, &r3[2], 4, hipMemcpyHostToDevice));
  }
  char rk[64]; snprintf(rk, sizeof(rk), "_rank_%03u_learner_", (unsigned)rank);
  const std::string fName = B + rk;
  FILE* fs = fopen((fName + "status.raw").c_str(), "r");
  FILE* fd = fopen((fName + "data.raw").c_str(), "rb");
  if (!fs || !fd) { if (fs) fclose(fs); if (fd) fclose(fd); return fail(h, HL_ERR_IO, "Learner status / data restart file " + fName + "*.raw not found"); }
  unsigned long nEps = 0, nObs = 0, seenE = 0, seenO = 0; long nInit = 0, doneGrad = 0; double Cmax = 0, beta = 0;
  int pass = 1;
  pass = pass && 1 == fscanf(fs, "nStoredEps: %lu\n", &nEps);
  pass = pass && 1 == fscanf(fs, "nStoredObs: %lu\n", &nObs);
  pass = pass && 1 == fscanf(fs, "nLocalSeenEps: %lu\n", &seenE);
  pass = pass && 1 == fscanf(fs, "nLocalSeenObs: %lu\n", &seenO);
  pass = pass && 1 == fscanf(fs, "nInitialData: %ld\n", &nInit);
  pass = pass && 1 == fscanf(fs, "nGradSteps: %ld\n", &doneGrad);
  pass = pass && 1 == fscanf(fs, "CmaxReFER: %le\n", &Cmax);
  pass = pass && 1 == fscanf(fs, "beta: %le\n", &beta);
  fclose(fs);
  if (!pass || doneGrad < 0) { fclose(fd); return fail(h, HL_ERR_IO, "Mismatch in restarted file " + fName + "status.raw"); }
  // episodes: unpack, append (same path as fresh ones), then put the stored per-step fields back
  struct Stored { std::vector<float> f6; int N; };
  std::vector<Stored> stored; stored.reserve(nEps);
  const int tup = dS + 1 + dA + h->polDim;
  for (unsigned long i = 0; i < nEps; ++i) {
    unsigned long N = 0;
    if (fread(&N, sizeof(unsigned long), 1, fd) != 1 || N < 2) { fclose(fd); return fail(h, HL_ERR_IO, "Unable to find sequence in " + fName + "data.raw"); }
    std::vector<float> buf((size_t)hl_packed_episode_size(h, (int32_t)N));
    if (fread(buf.data(), 4, buf.size(), fd) != buf.size()) { fclose(fd); return fail(h, HL_ERR_IO, "Truncated " + fName + "data.raw"); }
    int rc = hl_append_packed_episode(h, buf.data(), (int64_t)buf.size()); if (rc) { fclose(fd); return rc; }
    Stored st; st.N = (int)N; st.f6.assign(buf.begin() + (size_t)N * tup, buf.begin() + (size_t)N * (tup + 6));
    stored.push_back(std::move(st));
    const char* cp = reinterpret_cast<const char*>(buf.data() + (size_t)N * (tup + 6)) + sizeof(bool) + 8;
    std::memcpy(&h->order.front().sampled, cp, 8); std::memcpy(&h->order.front().agentID, cp + 8, 8);
  }
  fclose(fd);
  int rc = flushPending(h); if (rc) return rc;            // tables, counters, insertion-time Retrace
  float* dst[6] = {h->rp.RET, h->rp.ADV, h->rp.V, h->rp.DQ, h->rp.IMPW, h->rp.DKL};
  for (size_t i = 0; i < stored.size(); ++i) {           // episode i of the file sits at position nEps-1-i
    const EpMeta& e = h->order[stored.size() - 1 - i];
    for (int k = 0; k < 6; ++k)
      HIPCK(hipMemcpyAsync(dst[k] + e.off, stored[i].f6.data() + (size_t)k * e.N, (size_t)e.N * 4, hipMemcpyHostToDevice, h->stream));
  }
  HIPCK(hipStreamSynchronize(h->stream));
  // counters and ReF-ER state, then Episode::updateCumulative for every episode (MemoryBuffer.cpp:266)
  h->nSeenEps = (long long)seenE; h->nSeenSteps = (long long)seenO; h->nGatheredB4Startup = nInit; h->nGradSteps = doneGrad;
  h->countsDirty = true;
  rc = flushPending(h); if (rc) return rc;
  DevScalars sc; rc = syncScalarsToHost(h, &sc); if (rc) return rc;
  sc.Cmax = Cmax; sc.Cinv = 1 / Cmax; sc.beta = beta; sc.nGradSteps = doneGrad;
  HIPCK(hipMemcpy(h->sc, &sc, sizeof(DevScalars), hipMemcpyHostToDevice));
  rc = runSweep(h, nullptr, (int)h->order.size(), 1, /*skipRetrace*/1); if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  if ((unsigned long)h->nTransitions != nObs) return fail(h, HL_ERR_IO, "nStoredObs of the status file does not match the data file");
  h->initialized = true;      // Learner::initializeLearner is skipped for a restarted learner (Learner.cpp:51-54)
  return HL_OK;
}

// ---- checkpoint in the reference's format (Network::save / restart, Network/Network.cpp:22-68) ----
static void packBlob(const hl_learner* h, const std::vector<float>& P, std::vector<float>& out) {
  out.clear();
  for (const auto& l : h->lay) {
    const float* W = P.data() + l.indW; const float* Bv = P.data() + l.indB;
    if (l.type == 1) {
      for (int i = 0; i < l.nIn; ++i) for (int o = 0; o < l.size; ++o) out.push_back(W[o + (long long)l.ld * i]);
      for (int o = 0; o < l.size; ++o) out.push_back(Bv[o]);
    } else if (l.type == 2) {
      for (int o = 0; o < l.size; ++o) out.push_back(W[o]);
      for (int o = 0; o < l.size; ++o) out.push_back(Bv[o]);
    } else if (l.type == 4 || l.type == 5) {     // LSTMLayer::save / MGULayer::save (Layer_LSTM.h:186-197, Layer_GRU.h:248-258): weights, then biases, as they lie
      for (long long w = 0; w < (long long)l.ld * (l.nIn + l.size); ++w) out.push_back(W[w]);
      for (int o = 0; o < l.ld; ++o) out.push_back(Bv[o]);
    } else if (l.type == 6) {                    // Conv2DLayer::save (Layer_Conv2D.h:215-231): filters, then biases, as they lie
      for (int w = 0; w < l.nIn; ++w) out.push_back(W[w]);
      for (int o = 0; o < l.size; ++o) out.push_back(Bv[o]);
    } else for (int o = 0; o < l.size; ++o) out.push_back(Bv[o]);
  }
}
static void unpackBlob(const hl_learner* h, const std::vector<float>& in, std::vector<float>& P) {
  size_t k = 0;
  for (const auto& l : h->lay) {
    float* W = P.data() + l.indW; float* Bv = P.data() + l.indB;
    if (l.type == 1) {
      for (int i = 0; i < l.nIn; ++i) for (int o = 0; o < l.size; ++o) W[o + (long long)l.ld * i] = in[k++];
      for (int o = 0; o < l.size; ++o) Bv[o] = in[k++];
    } else if (l.type == 2) {
      for (int o = 0; o < l.size; ++o) W[o] = in[k++];
      for (int o = 0; o < l.size; ++o) Bv[o] = in[k++];
    } else if (l.type == 4 || l.type == 5) {
      for (long long w = 0; w < (long long)l.ld * (l.nIn + l.size); ++w) W[w] = in[k++];
      for (int o = 0; o < l.ld; ++o) Bv[o] = in[k++];
    } else if (l.type == 6) {
      for (int w = 0; w < l.nIn; ++w) W[w] = in[k++];
      for (int o = 0; o < l.size; ++o) Bv[o] = in[k++];
    } else for (int o = 0; o < l.size; ++o) Bv[o] = in[k++];
  }
}
int hl_save(hl_learner* h, const char* base) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  std::vector<float> P[3]; for (auto& v : P) v.resize((size_t)h->nParams);
  int rc = hl_get_params(h, P[0].data(), P[1].data(), P[2].data()); if (rc) return rc;
  const char* suf[3] = {"_weights", "_1stMom", "_2ndMom"};
  std::vector<float> buf;
  for (int b = 0; b < 3; ++b) {
    packBlob(h, P[b], buf);
    // like Network::save: write <name>_backup.raw first, then copy it over <name>.raw
    const std::string name = std::string(base) + suf[b] + ".raw", back = std::string(base) + suf[b] + "_backup.raw";
    for (const std::string& fn : {back, name}) {
      FILE* f = fopen(fn.c_str(), "wb");
      if (!f) return fail(h, HL_ERR_IO, "Unable to save into file " + fn);
      const size_t w = fwrite(buf.data(), sizeof(float), buf.size(), f);
      fclose(f);
      if (w != buf.size()) return fail(h, HL_ERR_IO, "short write to " + fn);
    }
  }
  return HL_OK;
}
int hl_restart(hl_learner* h, const char* base) {
  if (!h || !base) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  std::vector<float> P[3]; for (auto& v : P) v.resize((size_t)h->nParams);
  int rc = hl_get_params(h, P[0].data(), P[1].data(), P[2].data()); if (rc) return rc;
  size_t n = 0;
  for (const auto& l : h->lay) n += l.type == 1 ? (size_t)l.size * (l.nIn + 1) : (l.type == 2 ? 2 * (size_t)l.size :
                                  (l.type == 4 || l.type == 5 ? (size_t)l.ld * (l.nIn + l.size + 1) :
                                   (l.type == 6 ? (size_t)l.nIn + l.size : (size_t)l.size)));
  const char* suf[3] = {"_weights", "_1stMom", "_2ndMom"};
  for (int b = 0; b < 3; ++b) {
    const std::string name = std::string(base) + suf[b] + ".raw";
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) { if (b == 0) return fail(h, HL_ERR_IO, "Parameters restart file " + name + " not found."); continue; }
    std::vector<float> buf(n + 1);
    const size_t got = fread(buf.data(), sizeof(float), n + 1, f); fclose(f);
    if (got != n) return fail(h, HL_ERR_IO, "Mismatch in restarted file " + name);
    buf.resize(n); unpackBlob(h, buf, P[b]);
  }
  return hl_set_params(h, P[0].data(), P[1].data(), P[2].data());
}

// rollout inference: Approximator::forward(agent) for n states (RACER.cpp:30-59)
// pinned, device-mapped staging of rollout inference: outputs [ACT_MAXROWS][nOut] f64 | states f32 | completion stamps
static size_t actPinFloats(const hl_learner* h) { return std::max((size_t)ACT_MAXROWS * h->dIn, (size_t)(std::max(h->recWin, 1) + h->nApp) * h->dS); }
static int actPinEnsure(hl_learner* h) {
  if (h->actPin) return HL_OK;
  const size_t bytes = (size_t)ACT_MAXROWS * (h->nOut * sizeof(double) + sizeof(unsigned)) + actPinFloats(h) * sizeof(float) + 256;
  HIPCK(hipHostMalloc(reinterpret_cast<void**>(&h->actPin), bytes, hipHostMallocMapped));
  std::memset(h->actPin, 0, bytes);
  return HL_OK;
}
// the kernel stamps a row once its outputs are in host memory: poll the stamps (a stream synchronisation costs ~10 us more),
// give up after 2 s and fall back to it
static int actWait(hl_learner* h, volatile unsigned* pDone, int n, unsigned tag) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < n; ++r)
    while (pDone[r] != tag) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCK(hipStreamSynchronize(h->stream)); break; }
    }
  std::atomic_thread_fence(std::memory_order_acquire);
  return HL_OK;
}
int hl_forward(hl_learner* h, int32_t n, const float* states, double* outputs) {
  if (!h || n < 0 || (n > 0 && (!states || !outputs))) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (h->inStep) return fail(h, HL_ERR_STATE, "hl_forward between hl_step_begin and hl_step_end");
  if (h->recurrent) return fail(h, HL_ERR_UNSUPPORTED, "forward of a recurrent net needs the agent's history");
  // a few agents, dense network: one kernel, states and outputs through pinned host memory (misc.hip: act_forward_kernel)
  if (n > 0 && n <= ACT_MAXROWS && h->nConv == 0 && h->dIn <= ACT_MAXW && h->actFastOk) {
    { int rc = actPinEnsure(h); if (rc) return rc; }
    double* pOut = reinterpret_cast<double*>(h->actPin);
    float* pIn = reinterpret_cast<float*>(pOut + (size_t)ACT_MAXROWS * h->nOut);
    volatile unsigned* pDone = reinterpret_cast<volatile unsigned*>(pIn + actPinFloats(h));
    std::memcpy(pIn, states, (size_t)n * h->dIn * sizeof(float));
    ActArgs aa{}; aa.W = h->W; aa.stMean = h->rp.stMean; aa.stScale = h->rp.stScale; aa.in = pIn; aa.out = pOut; aa.done = pDone;
    aa.tag = ++h->actTag; if (aa.tag == 0) aa.tag = ++h->actTag;
    aa.dS = h->dS; aa.dIn = h->dIn; aa.nL = h->nHidden; aa.nDense = h->nDense; aa.nSig = h->nSig; aa.nOut = h->nOut; aa.ldWo = h->ldWo;
    aa.indWo = h->indWo; aa.indBo = h->indBo; aa.indBp = h->indBp; aa.outFunc = h->cfg.nnOutputFunc;
    for (int j = 0; j < h->nHidden; ++j) { const DevHidden& d = h->hid[j];
      aa.L[j] = ActLayer{d.nIn, d.size, d.ldW, d.func, d.hasRes, d.resW, d.indW, d.indB, d.indWr, d.indBr}; }
    HIPCK(launch_act_forward(aa, n, h->stream));
    { int rc = actWait(h, pDone, n, aa.tag); if (rc) return rc; }
    std::memcpy(outputs, pOut, (size_t)n * h->nOut * sizeof(double));
    return HL_OK;
  }
  { int rc = dropPresample(h); if (rc) return rc; }      // the forward pass borrows minibatch buffer 0
  // (with appended observations a row holds the raw state of step t followed by those of t-1 .. t-nAppendedObs)
  if (!h->dActS) { HIPCK(devAlloc(&h->dActS, (size_t)h->Mmax * h->dIn)); HIPCK(devAlloc(&h->dActO, (size_t)h->Mmax * h->nOut)); }
  const DevHidden& q = h->hid[h->nHidden - 1];
  for (int r0 = 0; r0 < n; r0 += h->Mmax) {
    const int m = std::min(h->Mmax, n - r0);
    HIPCK(hipMemcpyAsync(h->dActS, states + (size_t)r0 * h->dIn, (size_t)m * h->dIn * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(launch_act_standardize(h->sc, h->rp, h->dActS, m, h->dS, h->dIn, h->buf[0].X0, h->ldX0, h->stream));
    int rc = ensureConvPrep(h); if (rc) return rc;
    rc = launchForward(h, 0, h->stream, false, /*gather*/false); if (rc) return rc;
    HIPCK(launch_act_output(q.hasRes ? q.Rr : q.Y, q.ldA, q.size, h->W, h->indWo, h->indBo, h->indBp, h->ldWo, h->nDense, h->nSig, m,
                            h->dActO, h->stream, nullptr, 0, h->cfg.nnOutputFunc));
    HIPCK(hipMemcpyAsync(outputs + (size_t)r0 * h->nOut, h->dActO, (size_t)m * h->nOut * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
  }
  return HL_OK;
}

// the window kernels on the agent's last `win` states (`ctx` more in front of them for appended observations); a stack of two layer
// types as two launches, the lower segment's rows being the upper one's input
static int recActingForward(hl_learner* h, const float* dStates, int win, int ctx) {
  if (h->recSplit) {
    RecArgs lo = recArgs(h, 0, 0); lo.B = 1; lo.actStates = dStates; lo.actSteps = win; lo.actCtx = ctx;
    HIPCK(launch_rec_forward(lo, h->stream));
    RecArgs up = recArgs(h, 0, 1); up.B = 1; up.actStates = dStates; up.actSteps = win; up.actCtx = 0;
    HIPCK(launch_rec_forward(up, h->stream));
    return HL_OK;
  }
  RecArgs ra = recArgs(h, 0); ra.B = 1; ra.actStates = dStates; ra.actSteps = win; ra.actCtx = ctx;
  HIPCK(launch_rec_forward(ra, h->stream));
  return HL_OK;
}
int hl_forward_sequence(hl_learner* h, int32_t nSteps, const float* states, double* outputs) {
  if (!h || nSteps < 1 || !states || !outputs) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (!h->recurrent) {
    if (h->nApp == 0) return hl_forward(h, 1, states + (size_t)(nSteps - 1) * h->dS, outputs);
    // appended observations: the row hl_forward reads is the state of the last step followed by those of the steps before it
    // (Episode::standardizedState, Episode.h:172-183; steps before the first given one repeat it)
    std::vector<float> row((size_t)h->dIn);
    for (int j = 0; j <= h->nApp; ++j) { const int tt = std::max(nSteps - 1 - j, 0); std::memcpy(row.data() + (size_t)j * h->dS, states + (size_t)tt * h->dS, (size_t)h->dS * sizeof(float)); }
    return hl_forward(h, 1, row.data(), outputs);
  }
  if (h->inStep) return fail(h, HL_ERR_STATE, "hl_forward_sequence between hl_step_begin and hl_step_end");
  if (h->nConv > 0) {      // the window's stacked rows through the conv stack (as hl_forward does), then the window kernel on its rows
    if (nSteps > h->recWin + h->nApp) return fail(h, HL_ERR_BAD_ARG, "more steps than nnBPTTseq + 1 (+ nAppendedObs)");
    { int rc = dropPresample(h); if (rc) return rc; }
    const int win = std::min(nSteps, h->recWin), ctx = nSteps - win;
    std::vector<float> rows((size_t)win * h->dIn);
    for (int k = 0; k < win; ++k) for (int j = 0; j <= h->nApp; ++j) { const int g = std::max(ctx + k - j, 0);
      std::memcpy(rows.data() + (size_t)k * h->dIn + (size_t)j * h->dS, states + (size_t)g * h->dS, (size_t)h->dS * sizeof(float)); }
    if (!h->dActS) { HIPCK(devAlloc(&h->dActS, (size_t)h->convMmax * h->dIn)); HIPCK(devAlloc(&h->dActO, (size_t)h->Mmax * h->nOut)); }
    HIPCK(hipMemcpyAsync(h->dActS, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(launch_act_standardize(h->sc, h->rp, h->dActS, win, h->dS, h->dIn, h->buf[0].X0, h->ldX0, h->stream));
    int rc = ensureConvPrep(h); if (rc) return rc;
    rc = launchFront(h, 0, h->stream, /*gather*/false); if (rc) return rc;
    const DevHidden& q = h->hid[h->nHidden - 1];
    { const int rc2 = recActingForward(h, h->dActS, win, 0); if (rc2) return rc2; }      // (the rows come from Xin; the states only mark the call as acting)
    HIPCK(launch_act_output(q.hasRes ? q.Rr : q.Y, q.ldA, q.size, h->W, h->indWo, h->indBo, h->indBp, h->ldWo, h->nDense, h->nSig, 1,
                            h->dActO, h->stream, nullptr, 0, h->cfg.nnOutputFunc));
    HIPCK(hipMemcpyAsync(outputs, h->dActO, (size_t)h->nOut * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return HL_OK;
  }
  // (appended observations: up to nAppendedObs further states in front of the window, which only feed the window's first steps)
  if (nSteps > h->recWin + h->nApp) return fail(h, HL_ERR_BAD_ARG, "more steps than nnBPTTseq + 1 (+ nAppendedObs)");
  // states and outputs through pinned host memory, completion by stamp (as hl_forward): two launches, no staged copies
  { int rc = actPinEnsure(h); if (rc) return rc; }
  double* pOut = reinterpret_cast<double*>(h->actPin);
  float* pIn = reinterpret_cast<float*>(pOut + (size_t)ACT_MAXROWS * h->nOut);
  volatile unsigned* pDone = reinterpret_cast<volatile unsigned*>(pIn + actPinFloats(h));
  std::memcpy(pIn, states, (size_t)nSteps * h->dS * sizeof(float));
  unsigned tag = ++h->actTag; if (tag == 0) tag = ++h->actTag;
  const DevHidden& q = h->hid[h->nHidden - 1];
  { const int win = std::min(nSteps, h->recWin); const int rc2 = recActingForward(h, pIn, win, nSteps - win); if (rc2) return rc2; }
  HIPCK(launch_act_output(q.hasRes ? q.Rr : q.Y, q.ldA, q.size, h->W, h->indWo, h->indBo, h->indBp, h->ldWo, h->nDense, h->nSig, 1,
                          pOut, h->stream, const_cast<unsigned*>(pDone), tag, h->cfg.nnOutputFunc));
  { int rc = actWait(h, pDone, 1, tag); if (rc) return rc; }
  std::memcpy(outputs, pOut, (size_t)h->nOut * sizeof(double));
  return HL_OK;
}

int hl_sync(hl_learner* h) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK_RAW(h);
  if (h->tailNotify && h->notifyPin) {
    // the last thing queued is an exact-size graph: its final node stores the count of such graphs into pinned memory
    const unsigned want = h->notifyIssued;
    volatile unsigned* w = h->notifyPin;
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spin = 0;; ++spin) {
      if ((int)(*w - want) >= 0) return HL_OK;
      if ((spin & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;   // (a sweep or a stalled device: wait the ordinary way)
    }
  }
  h->tailNotify = false;
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}

// A caller that will step n gradient steps per call (the reference's task loop does one; bench.py --steps n) announces it:
// the graph of exactly n steps is captured here -- one-off costs belong to set-up, as in hl_initialize.
int hl_prepare_steps(hl_learner* h, int32_t n) {
  if (!h || n < 1) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (!h->initialized) return fail(h, HL_ERR_STATE, "hl_prepare_steps before hl_initialize");
  int rc = flushPending(h); if (rc) return rc;
  rc = prepareExact(h, n); if (rc) return rc;
  return touchReplay(h);
}

int hl_set_tap(hl_learner* h, int32_t) { return h ? HL_OK : HL_ERR_BAD_ARG; }   // taps are always recorded

int hl_readback(hl_learner* h, int32_t what, void* dst, int64_t bytes) {
  if (!h || !dst) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  const int B = h->B;
  HIPCK(hipStreamSynchronize(h->stream));
  const DevBatch& bt = h->buf[h->lastParity].bt;
  auto copy = [&](const void* src, int64_t n) -> int {
    if (bytes < n) return HL_ERR_BAD_ARG;
    HIPCK(hipMemcpy(dst, src, (size_t)n, hipMemcpyDeviceToHost)); return HL_OK;
  };
  switch (what) {
    case HL_TAP_FLAT: return copy(bt.flat, (int64_t)B * 8);
    case HL_TAP_TAG: return copy(bt.tag, (int64_t)B * 8);
    case HL_TAP_EPISODE: case HL_TAP_TSTEP: {
      if (bytes < (int64_t)B * 8) return HL_ERR_BAD_ARG;
      std::vector<int> tmp(B);
      HIPCK(hipMemcpy(tmp.data(), what == HL_TAP_TSTEP ? bt.t : bt.pos, B * sizeof(int), hipMemcpyDeviceToHost));
      int64_t* o = (int64_t*)dst;
      for (int b = 0; b < B; ++b) o[b] = tmp[b];
      return HL_OK;
    }
    case HL_TAP_STATE: {
      if (bytes < (int64_t)B * h->dIn * 4) return HL_ERR_BAD_ARG;
      if (h->recurrent || convFromReplay(h)) {   // recurrent layers and row-block convolutions read their windows straight from the replay: the rows are assembled on demand
        StackGatherArgs ga{}; ga.sc = h->sc; ga.rp = h->rp; ga.bt = bt; ga.B = B; ga.dS = h->dS; ga.nApp = h->nApp; ga.parity = h->lastParity;
        ga.X0 = h->buf[h->lastParity].X0; ga.ldX0 = h->ldX0;
        HIPCK(launch_stack_gather(ga, h->Mmax, h->stream)); HIPCK(hipStreamSynchronize(h->stream));
      }
      HIPCK(hipMemcpy2D(dst, (size_t)h->dIn * 4, h->buf[h->lastParity].X0, (size_t)h->ldX0 * 4, (size_t)h->dIn * 4, B, hipMemcpyDeviceToHost));
      return HL_OK;
    }
    case HL_TAP_OUTPUT: return copy(bt.O, (int64_t)B * h->nOut * 8);
    case HL_TAP_OUTGRAD: return copy(bt.G, (int64_t)B * h->nOut * 8);
    case HL_TAP_RHO: return copy(bt.rho, (int64_t)B * 8);
    case HL_TAP_DKL: return copy(bt.dkl, (int64_t)B * 8);
    case HL_TAP_DELTAQ: return copy(bt.dq, (int64_t)B * 8);
    case HL_TAP_FAR: return copy(bt.far, (int64_t)B);
    case HL_TAP_GRADSUM: return copy(h->G, (int64_t)h->nParams * 4);
  }
  return HL_ERR_BAD_ARG;
}

int hl_get_scalars(hl_learner* h, hl_scalars* o) {
  if (!h || !o) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = flushPending(h); if (rc) return rc;
  DevScalars s; rc = syncScalarsToHost(h, &s); if (rc) return rc;
  o->beta = s.beta; o->alpha = s.alpha; o->CmaxRet = s.Cmax; o->CinvRet = s.Cinv;
  o->nGradSteps = s.nGradSteps; o->nStoredSteps = h->nTransitions; o->nStoredEps = (int64_t)h->order.size();
  o->nFarPolicySteps = s.nFarStat; o->nSeenSteps = s.seenUpd[1]; o->nSeenEps = s.seenUpd[0];
  o->adam_beta_t_1 = s.adam_bt1; o->adam_beta_t_2 = s.adam_bt2; o->adam_nStep = s.nStep;
  return HL_OK;
}
int hl_get_counts(hl_learner* h, int64_t* nStoredSteps, int64_t* nStoredEps, int64_t* nGradSteps, int64_t* nSeenSteps, int64_t* nSeenEps) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (nStoredSteps) *nStoredSteps = h->nTransitions;
  if (nStoredEps) *nStoredEps = (int64_t)h->order.size();
  if (nGradSteps) *nGradSteps = h->nGradSteps;
  if (nSeenSteps) *nSeenSteps = h->nSeenSteps;
  if (nSeenEps) *nSeenEps = h->nSeenEps;
  return HL_OK;
}
int hl_get_initial_data(hl_learner* h, int64_t* n) {
  if (!h || !n) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  *n = (int64_t)h->nGatheredB4Startup;
  return HL_OK;
}
int hl_get_stats(hl_learner* h, hl_stats* o) {
  if (!h || !o) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = flushPending(h); if (rc) return rc;
  // ReplayStats as of the last step's statistics pass (MemoryProcessing::updateTrainingStatistics): episodes that arrived since do
  // not enter yet -- the snapshot taken before they were ingested (refreshInsertionStats) is that state; with no arrivals since
  // the last step the aggregates the device holds now are
  const double* src = h->dStatsOut;
  if (h->statsFresh && h->anyStep) src = h->dStatsIns;
  else HIPCK(launch_stats(h->sc, h->rp, (int)h->order.size(), h->dStatsOut, h->stream));
  double out[16];
  HIPCK(hipMemcpyAsync(out, src, sizeof(out), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  o->avgKLdivergence = out[0]; o->avgSquaredErr = out[1]; o->maxAbsError = out[2]; o->avgReturn = out[3];
  o->avgQ = out[4]; o->stdevQ = out[5]; o->minQ = out[6]; o->maxQ = out[7]; o->nFarPolicySteps = (int64_t)out[8];
  DevScalars sc; rc = syncScalarsToHost(h, &sc); if (rc) return rc;      // (live counters: a snapshot may predate the last statistics line)
  o->countReturnsEstimateUpdates = (int64_t)sc.cntRetUpd; o->sumReturnsEstimateErrors = sc.sumRetErr;
  return HL_OK;
}

// ---- RCCL over xGMI (C1-C4 of SURVEY.md 2.4) -------------------------------------------------------
int hl_comm_unique_id(uint8_t id[128]) {
  if (!id) return HL_ERR_BAD_ARG;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return HL_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) <= 128, "unique id does not fit");
  std::memset(id, 0, 128); std::memcpy(id, &u, sizeof(u));
  return HL_OK;
}
int hl_comm_init(hl_learner* h, const uint8_t id[128]) {
  if (!h || !id) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
  HIPCK(hipSetDevice(h->dev));
  NCCLCK(ncclCommInitRank(&h->comm, h->cfg.n_ranks, u, h->cfg.rank));
  // identical initial weights on every replica: MPI_Bcast from rank 0 (Network/Builder.cpp:143-144)
  NCCLCK(ncclBroadcast(h->W, h->W, (size_t)h->nParams, ncclFloat, 0, h->comm, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return HL_OK;
}

// ---- the same sums without RCCL: peer-mapped windows, one kernel per collective (xchg.hip) --------
// handle: [0,64) hipIpcMemHandle_t | [64,72) the window's address in the exporting process | [72,76) its pid | [76,80) its
// device | [80,88) window bytes
namespace {
size_t xchgMsgBytes(const hl_learner* h) {
  size_t b = ((size_t)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS) * sizeof(float);
  b = std::max(b, (size_t)(2 * h->dS + 3) * sizeof(double));
  return (std::max(b, (size_t)64) + 255) & ~(size_t)255;
}
}  // namespace
int hl_xchg_export(hl_learner* h, uint8_t out[HL_XCHG_HANDLE_BYTES]) {
  if (!h || !out) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  if (h->cfg.n_ranks < 2 || h->cfg.n_ranks > XCHG_MAX_RANKS) return fail(h, HL_ERR_BAD_ARG, "hl_xchg_export: 2..16 replicas");
  HIPCK(hipSetDevice(h->dev));
  auto& x = h->xchg;
  if (!x.win) {
    const size_t R = (size_t)h->cfg.n_ranks;
    x.slotBytes = xchgMsgBytes(h);
    x.slotsOffset = (2 * R * XCHG_CHUNKS * sizeof(unsigned long long) + 255) & ~(size_t)255;
    x.winBytes = x.slotsOffset + 2 * R * x.slotBytes;
    // uncached: the peers' stores land in HBM behind this device's L2, the owner's loads must not be served from it
    x.win = windowPoolGet(h->dev, x.winBytes);
    if (!x.win) HIPCK(hipExtMallocWithFlags(reinterpret_cast<void**>(&x.win), x.winBytes, hipDeviceMallocUncached));
    HIPCK(hipMemset(x.win, 0, x.winBytes));
    HIPCK(devAlloc(&x.ctl, 1));
    HIPCK(devAlloc(&x.dPeers, R));
    HIPCK(hipDeviceSynchronize());
  }
  std::memset(out, 0, HL_XCHG_HANDLE_BYTES);
  hipIpcMemHandle_t hd;
  if (hipIpcGetMemHandle(&hd, x.win) == hipSuccess) std::memcpy(out, &hd, sizeof(hd));
  else (void)hipGetLastError();              // (same-process peers do not need it; others fail in hl_xchg_connect)
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "ipc handle does not fit");
  const unsigned long long addr = (unsigned long long)(uintptr_t)x.win, bytes = x.winBytes;
  const int pid = (int)getpid(), dev = h->dev;
  std::memcpy(out + 64, &addr, 8); std::memcpy(out + 72, &pid, 4); std::memcpy(out + 76, &dev, 4); std::memcpy(out + 80, &bytes, 8);
  return HL_OK;
}
int hl_xchg_connect(hl_learner* h, const uint8_t* handles) {
  if (!h || !handles) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  auto& x = h->xchg;
  if (!x.win) return fail(h, HL_ERR_BAD_ARG, "hl_xchg_connect before hl_xchg_export");
  HIPCK(hipSetDevice(h->dev));
  const int R = h->cfg.n_ranks;
  std::vector<unsigned char*> peers((size_t)R, nullptr);
  for (int r = 0; r < R; ++r) {
    const uint8_t* e = handles + (size_t)r * HL_XCHG_HANDLE_BYTES;
    unsigned long long addr, bytes; int pid, dev;
    std::memcpy(&addr, e + 64, 8); std::memcpy(&pid, e + 72, 4); std::memcpy(&dev, e + 76, 4); std::memcpy(&bytes, e + 80, 8);
    if (bytes != x.winBytes) return fail(h, HL_ERR_BAD_ARG, "hl_xchg_connect: the replicas' windows differ in size (different networks?)");
    if (r == h->cfg.rank) {
      if (addr != (unsigned long long)(uintptr_t)x.win || pid != (int)getpid()) return fail(h, HL_ERR_BAD_ARG, "hl_xchg_connect: entry [rank] is not this learner's handle");
      peers[(size_t)r] = x.win;
    } else if (pid == (int)getpid()) {        // a learner of this process: its pointer as it is
      if (dev != h->dev) {
        const hipError_t pe = hipDeviceEnablePeerAccess(dev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return hipFail(h, pe, "hipDeviceEnablePeerAccess");
        (void)hipGetLastError();
      }
      peers[(size_t)r] = reinterpret_cast<unsigned char*>((uintptr_t)addr);
    } else {
      hipIpcMemHandle_t hd; std::memcpy(&hd, e, sizeof(hd));
      void* q = nullptr;
      HIPCK(hipIpcOpenMemHandle(&q, hd, hipIpcMemLazyEnablePeerAccess));
      x.opened.push_back(q);
      peers[(size_t)r] = static_cast<unsigned char*>(q);
    }
  }
  HIPCK(hipMemcpy(x.dPeers, peers.data(), (size_t)R * sizeof(unsigned char*), hipMemcpyHostToDevice));
  // Replicas that SHARE a device (the one-GPU test box: 2 - 8 of them; never on a node, one process per GPU) wait for each other inside
  // their kernels while competing for the same CUs: 8 x 64 waiting chunk workgroups of the folded weight-gradient launch (each with that
  // launch's registers and LDS) kept the peers' fused kernels from getting their panel groups resident -- bounded spins, device error
  // 77.  The message is therefore cut into fewer chunks the more replicas sit on the busiest device; the cut is part of the wire
  // protocol and every replica derives the same figure from the same handles.
  { int most = 1;
    for (int r = 0; r < R; ++r) {
      int devR, same = 0; std::memcpy(&devR, handles + (size_t)r * HL_XCHG_HANDLE_BYTES + 76, 4);
      for (int q = 0; q < R; ++q) { int devQ; std::memcpy(&devQ, handles + (size_t)q * HL_XCHG_HANDLE_BYTES + 76, 4); same += devQ == devR ? 1 : 0; }
      most = std::max(most, same);
    }
    x.maxChunks = most <= 1 ? XCHG_CHUNKS : std::max(4, XCHG_CHUNKS / most); }
  x.on = true;
  h->graphsStale = true;
  // identical initial weights on every replica: MPI_Bcast from rank 0 (Network/Builder.cpp:143-144) as a sum with zeros
  if (h->cfg.rank == 0) HIPCK(hipMemcpyAsync(h->G, h->W, (size_t)h->nParams * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  else HIPCK(hipMemsetAsync(h->G, 0, (size_t)h->nParams * sizeof(float), h->stream));
  // dense networks: every gradient element comes out of a tile of the weight-gradient launch, which then stores it into the peers'
  // windows itself (recurrent and convolutional nets have further gradient producers -- split-row joins, filter gradients: the
  // exchange kernel keeps pushing their message)
  { const char* np = getenv("SMARTIES_HIP_NO_PUSH"); h->pushOk = !(np && np[0] == '1') && !h->recurrent && h->nConv == 0 && !h->bigBatch; }      // (local batches above 1024: split-row joins and the 64 x 64 tiles never push -- the exchange kernel sends their gradient)
  // The exchange folded into the weight-gradient launch (two launches per replica step instead of three: xchg_dev.h, dw_table_kernel) is
  // OFF unless SMARTIES_HIP_FOLD=1.  Built and measured in round 6: bit-equal to the host-formed sums in every fresh process, no faster
  // than the three-launch step (33.6 us either way, tools/replica_loopback.py: its chunk workgroups wait for the bookkeeping rider) --
  // and it hands gradient tiles from the producing workgroups to the summing ones INSIDE one launch, across XCDs, on the strength of
  // acknowledged window stores alone.  In a process that had created and destroyed other learners before (recycled device memory)
  // that hand-off delivered stale bytes in 1 of 4 runs of the 8-replica tests (1 of 18 with system-scope loads; 0 with a system-scope
  // fence per tile, which costs 15 us per step).  The three-launch step hands over at kernel boundaries only.
  { const char* fo = getenv("SMARTIES_HIP_FOLD"); h->foldOk = h->pushOk && fo && fo[0] == '1'; }
  int rc = xchgAllreduce(h, h->G, (size_t)h->nParams, 0); if (rc) return rc;
  HIPCK(hipMemcpyAsync(h->W, h->G, (size_t)h->nParams * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  DevScalars sc; rc = syncScalarsToHost(h, &sc); if (rc) return rc;       // (a peer that never showed up: the wait timed out)
  return HL_OK;
}

// ---- timing taps (HIP events on the library's stream) ---------------------------------------------
int hl_timing_enable(hl_learner* h, int32_t e) {
  if (!h) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  timerFlush(h);
  h->timing = e != 0;
  if (h->timing) { std::fill(h->tsum.begin(), h->tsum.end(), 0.0); std::fill(h->tcnt.begin(), h->tcnt.end(), 0); }
  return HL_OK;
}
int hl_timing_get(hl_learner* h, const char* kernel, double* avg_ms, int64_t* launches) {
  if (!h || !kernel) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  timerFlush(h);
  for (size_t i = 0; i < h->tnames.size(); ++i) if (h->tnames[i] == kernel) {
    if (avg_ms) *avg_ms = h->tcnt[i] ? h->tsum[i] / h->tcnt[i] : 0.0;
    if (launches) *launches = h->tcnt[i];
    return HL_OK;
  }
  if (avg_ms) *avg_ms = 0;
  if (launches) *launches = 0;
  return HL_OK;
}

}  // extern "C"

// ---- development aid: wall-clock time of ONE kernel of the step, replayed `reps` times from a graph
//      (which: 0 sample, 1 fwd0, 2 fwd(last), 3 head, 4 dx(last), 5 dw+adam, 6 post, 7 whole overlapped step) ----
extern "C" HL_API int hl_debug_kernel_time(hl_learner* h, int which, int reps, int variant, double* us_per_launch) {
  if (!h || !us_per_launch || reps <= 0) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  int rc = flushPending(h); if (rc) return rc;
  rc = dropPresample(h); if (rc) return rc;
  if (h->cfg.dataSamplingAlgo != HL_SAMPLE_UNIFORM) return fail(h, HL_ERR_UNSUPPORTED, "kernel profiles replay captured launches: not with the prioritised samplers (their table is rebuilt per minibatch)");
  if ((which == 1 || which == 21) && (h->buf[0].fwdIdx.empty() || h->buf[0].fwdIdx[0] < 0)) return fail(h, HL_ERR_UNSUPPORTED, "no dense first layer to profile (convolutional preprocessing)");
  if ((which == 2 || which == 22 || which == 4 || which == 24) && (h->recurrent || h->buf[0].fwdIdx.empty())) return fail(h, HL_ERR_UNSUPPORTED, "dense-layer profiles do not apply to recurrent networks");
  h->dbgVariant = variant;
  GraphSlot slot;
  if (which == 7) {
    rc = launchSample(h, 0, nullptr, true, h->stream); if (rc) return rc;
    rc = captureSteps(h, reps & ~1, 0, &slot); if (rc) { h->dbgVariant = 0; return rc; }   // (even: every replay starts with buffer 0)
  } else {
    const AdamHyper hyp = adamHyper(h, 0);
    const StepBuf& sb = h->buf[0];
    HIPCK(hipStreamSynchronize(h->stream));
    HIPCK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < reps && !rc; ++r) {
      hipError_t e = hipSuccess;
      switch (which) {
        case 0: rc = launchSample(h, 0, nullptr, true, h->stream); break;
        case 1: e = launch_gemm(GEMM_ROLE_FWD0, h->dProbs + sb.fwdIdx[0], 1, sb.fwdBlocks[0], h->sc, hyp, nullptr, h->stream); break;
        case 2: e = launch_gemm(GEMM_ROLE_FWD, h->dProbs + sb.fwdIdx[h->nHidden - 1], 1, sb.fwdBlocks[h->nHidden - 1], h->sc, hyp, nullptr, h->stream); break;
        case 3: rc = launchHead(h, 0, h->stream); break;
        case 4: if (!sb.dxIdx.empty()) e = launch_gemm(GEMM_ROLE_DX, h->dProbs + sb.dxIdx[0], 1, sb.dxBlocks[0], h->sc, hyp, nullptr, h->stream); break;
        case 5: e = launch_gemm(GEMM_ROLE_DW, h->dProbs + sb.dwAdamIdx, sb.dwCount, sb.dwBlocks, h->sc, hyp, nullptr, h->stream); break;
        case 6: rc = launchPost(h, 0, POST_AGG, h->stream); break;
        case 8: case 9: case 10: { const SampleArgs sa = sampleArgs(h, 0, nullptr, false);
          e = launch_step_tail(nullptr, &sa, h->stream, which == 8 ? PH_A : which == 9 ? PH_B : PH_C); break; }
        case 11: rc = launchPost(h, 0, POST_AGG | POST_BETA, h->stream); break;
        case 12: e = launch_empty(h->stream); break;
        // 21..25: the five launches of a replayed step exactly as captureSteps issues them (with riders)
        case 21: { ExtraArgs ex = extraSample(h, 1, h->nHidden == 1 ? (PH_A | PH_B) : PH_A);
          e = launch_gemm(GEMM_ROLE_FWD0, h->dProbs + sb.fwdIdx[0], 1, sb.fwdBlocks[0], h->sc, hyp, &ex, h->stream); break; }
        case 22: { ExtraArgs ex = extraSample(h, 1, PH_B); const int j = h->nHidden - 1;
          e = launch_gemm(GEMM_ROLE_FWD, h->dProbs + sb.fwdIdx[j], 1, sb.fwdBlocks[j], h->sc, hyp, &ex, h->stream); break; }
        case 23: rc = launchHead(h, 0, h->stream, true); break;
        case 24: if (!sb.dxIdx.empty()) { ExtraArgs ex{}; ex.role = 2; ex.post = postArgs(h, 0, POST_AGG | POST_BETA);
          e = launch_gemm(GEMM_ROLE_DX, h->dProbs + sb.dxIdx[0], 1, sb.dxBlocks[0], h->sc, hyp, &ex, h->stream); } break;
        case 25: e = launch_gemm(GEMM_ROLE_DW, h->dProbs + sb.dwAdamIdx, sb.dwCount, sb.dwBlocks, h->sc, hyp, nullptr, h->stream); break;
        // fused path: 26 = forward+head+dX (+ sampler phases A,B), 27 = dW+Adam (+ phase C, bookkeeping),
        // 28 / 29 = the same two kernels without riders
        case 26: rc = h->fusedOk ? launchFused(h, 0, h->stream, true) : fail(h, HL_ERR_UNSUPPORTED, "fused kernel not used for this network"); break;
        case 27: rc = h->fusedOk ? launchWeightGrad(h, 0, true, h->stream, true, true) : fail(h, HL_ERR_UNSUPPORTED, "fused kernel not used for this network"); break;
        case 28: rc = h->fusedOk ? launchFused(h, 0, h->stream, false) : fail(h, HL_ERR_UNSUPPORTED, "fused kernel not used for this network"); break;
        case 29: rc = h->fusedOk ? launchWeightGrad(h, 0, true, h->stream, false, false) : fail(h, HL_ERR_UNSUPPORTED, "fused kernel not used for this network"); break;
        default: break;
      }
      if (e != hipSuccess) rc = hipFail(h, e, "debug launch");
    }
    hipError_t e = hipStreamEndCapture(h->stream, &slot.graph);
    if (!rc && e != hipSuccess) rc = hipFail(h, e, "hipStreamEndCapture");
    if (!rc && hipGraphInstantiate(&slot.exec, slot.graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(h, HL_ERR_HIP, "instantiate");
    if (rc) { h->dbgVariant = 0; return rc; }
  }
  h->dbgVariant = 0;
  HIPCK(hipGraphLaunch(slot.exec, h->stream)); HIPCK(hipStreamSynchronize(h->stream));
  const int iters = 20;
  hipEvent_t ev0, ev1;
  HIPCK(hipEventCreate(&ev0)); HIPCK(hipEventCreate(&ev1));
  HIPCK(hipEventRecord(ev0, h->stream));
  for (int i = 0; i < iters; ++i) HIPCK(hipGraphLaunch(slot.exec, h->stream));
  HIPCK(hipEventRecord(ev1, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  float ms = 0.f; HIPCK(hipEventElapsedTime(&ms, ev0, ev1));
  hipEventDestroy(ev0); hipEventDestroy(ev1);
  *us_per_launch = (double)ms * 1e3 / ((double)iters * (which == 7 ? (reps & ~1) : reps));
  hipGraphExecDestroy(slot.exec); hipGraphDestroy(slot.graph);
  if (which == 7) h->nGradSteps += (long long)(iters + 1) * (reps & ~1);
  return HL_OK;
}

extern "C" HL_API int hl_kernel_profile(hl_learner* h, int which, int reps, double* us_per_launch) {
  return hl_debug_kernel_time(h, which, reps, 0, us_per_launch);
}

// RCCL calls issued or captured so far (tests: eager and replayed steps speak the same wire protocol)
extern "C" HL_API int64_t hl_debug_collectives(const hl_learner* h) { return h ? h->nCollectives : -1; }
// kernel nodes of the replayed graph of `steps` plain steps (one of GRAPH_SIZES; captured on demand): how many launches a step is made of
// (tests: a folded replica step = 2 kernels, the round-5 replica step = 3; development API like hl_debug_collectives, not in the header)
extern "C" HL_API int64_t hl_debug_graph_kernels(hl_learner* h, int32_t steps) {
  if (!h) return -1;
  HL_LOCK(h);
  constexpr int NS = (int)(sizeof(GRAPH_SIZES) / sizeof(GRAPH_SIZES[0]));
  if (h->graphsStale) { invalidateGraphs(h); h->graphsStale = false; }
  if (captureAllGraphs(h) != HL_OK) return -1;
  for (int j = 0; j < NS; ++j) if (GRAPH_SIZES[j] == steps && h->graphs[j][0].graph) {
    size_t n = 0;
    if (hipGraphGetNodes(h->graphs[j][0].graph, nullptr, &n) != hipSuccess) return -1;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(h->graphs[j][0].graph, nodes.data(), &n) != hipSuccess) return -1;
    int64_t k = 0;
    for (size_t i = 0; i < n; ++i) { hipGraphNodeType t; if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k; }
    return k;
  }
  return -1;
}
// fused kernel: -1 not in use, 0 panel exchange through the shared L2 (probe: workgroup b on XCD b % 8), 1 through agent-scope accesses
extern "C" HL_API int hl_debug_panel_mode(const hl_learner* h) { return !h || !h->fusedOk ? -1 : (h->xcdSafe ? 1 : 0); }

// the prioritised samplers' tables as the last step built them (tests: sequential normalisation / partial_sum)
extern "C" HL_API int64_t hl_debug_per_table(hl_learner* h, float* prob, double* cp, int64_t cap) {
  if (!h) return -1;
  HL_LOCK(h);
  if (!h->perProb) return 0;
  const int64_t n = h->cfg.dataSamplingAlgo == HL_SAMPLE_PERSEQ ? (int64_t)h->order.size() : (int64_t)h->nTransitions;
  if (n > cap) return -n;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
  if (prob && hipMemcpy(prob, h->perProb, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (cp && hipMemcpy(cp, h->perCp, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}
// the discrete distribution's cumulative table of n host probabilities by per.hip's scan (which = 0: the grid form where the table
// is long enough, 2: one workgroup) or its sequential walk (which = 1); returns the milliseconds of the launches (HIP events),
// negative on failure
extern "C" HL_API double hl_debug_per_scan(const float* prob, double* cp, int64_t n, int which) {
  if (!prob || !cp || n < 2) return -1;
  float* dP = nullptr; double* dC = nullptr; void* dS = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; float ms = -1;
  bool ok = hipMalloc(&dP, n * sizeof(float)) == hipSuccess && hipMalloc(&dC, n * sizeof(double)) == hipSuccess && hipMalloc(&dS, per_scan_scratch_bytes(n)) == hipSuccess
            && hipMemcpy(dP, prob, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess && hipMemset(dC, 0, n * sizeof(double)) == hipSuccess
            && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  if (ok) ok = launch_per_scan(dP, dC, n, which, dS, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess;      // (warm)
  if (ok) ok = hipMemset(dC, 0, n * sizeof(double)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (ok) ok = hipEventRecord(e0, nullptr) == hipSuccess && launch_per_scan(dP, dC, n, which, dS, nullptr) == hipSuccess && hipEventRecord(e1, nullptr) == hipSuccess
               && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess
               && hipMemcpy(cp, dC, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); if (dP) hipFree(dP); if (dC) hipFree(dC); if (dS) hipFree(dS);
  return ok ? (double)ms : -1.0;
}
extern "C" HL_API int hl_debug_step_stamps(hl_learner* h, long long out[128]) {      // (library built with -DHL_STEP_STAMPS)
  if (!h || !out) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  DevScalars s; int rc = syncScalarsToHost(h, &s); if (rc) return rc;
  std::memcpy(out, s.dbgStep, sizeof(s.dbgStep));
  return HL_OK;
}
extern "C" HL_API int hl_debug_stamps(hl_learner* h, long long out[32]) {
  if (!h || !out) return HL_ERR_BAD_ARG;
  HL_LOCK(h);
  DevScalars s; int rc = syncScalarsToHost(h, &s); if (rc) return rc;
  std::memcpy(out, s.dbgT, sizeof(s.dbgT));
  return HL_OK;
}
