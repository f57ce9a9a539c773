// smarties_amd/csrc/kernels.h -- launchers of the gfx950 kernels (kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <utility>
#include "hl_types.h"

namespace hl {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: raise it once per (device, kernel)
// and size, from whatever thread launches first (learners on different devices may live in one process)
inline hipError_t ensureDynLds(const void* kernel, size_t bytes) {
  static std::mutex m; static std::map<std::pair<int, const void*>, size_t> done;
  int dev = 0; (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(m);
  size_t& have = done[std::make_pair(dev, kernel)];
  if (bytes <= have) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

struct SampleArgs {
  DevScalars* sc; DevReplay rp; DevBatch bt;
  int B, dS, ldX0;        // local batch, state dim, leading dim of X0
  float* X0;              // [Mmax][ldX0] standardized states (MiniBatch::S)
  const long long* flatGiven;  // != nullptr: use these indices instead of drawing
  int adamDraws;          // mt19937 draws consumed by the Adam step (Optimizer.cpp:139)
  int parity;             // minibatch buffer written (bt / X0 passed here belong to it)
  int computeEta;         // also derive DevScalars::etaEff[parity] (first step of a launch sequence)
  int perAlgo;            // HL_SAMPLE_*: the prioritised samplers draw through the cumulative table `perCp` (per.hip)
  const double* perCp; long long perN;   // cumulative probabilities of the perN transitions (PERrank, PERerr) or episodes (PERseq); perN < 2: always 0
  int selfSearch;         // gather helpers (gatherHelper) run the index -> (episode, step) search themselves instead of waiting for the rider's hand-off
  int tagSeq;             // gather hand-off tag: 0 = nStep + 1 (nothing in the publishing kernel changes nStep), 1 = sampleSeq (the dW kernel:
                          // its bookkeeping rider advances nStep while the sampler's phase C and the gather helpers run)
  int noGather;           // phase C stops after the index -> (episode, step) search: the states are gathered by
                          // stack_gather_kernel (appended observations / convolutional input, conv.hip)
  int backupRng;          // keep the generator state as of before the draws in DevScalars::rngBak (pre-sampling riders)
  float eta0; double epsAnneal;
};

struct HeadArgs {
  DevScalars* sc; DevReplay rp; DevBatch bt;
  int B, dA, nDense, nOut, H;       // H = width of the last hidden block
  int deferBeta;                    // 1 (gemm16.hip: step_chain_kernel only): beta of this step comes from a rider of the same launch, the heads wait for its number
  int nAdv;                         // advantage outputs between V and the policy mean: 0 (VRACER) or 1 + 2 dA (Gaussian)
  int nOpt, nSig;                   // discrete head: number of options (else 0); size of the sigma ParamLayer (dA or 0)
  const float* Yin; int ldY;        // input of the output layer [Mmax][ldY]
  const float* Xlast; const float* Ylast; int func;   // last hidden block pre/post activation (for act')
  const float* params;              // weight blob
  long long indWo, indBo, indBp; int ldWo;   // output dense W/b, ParamLayer bias
  float* dOut; int ldDo;            // [B][ldDo] output-layer deltas (f32)
  float* Dres; float* D; int ldD;   // gradient wrt last hidden block output / after act'
  unsigned char bounded[HL_MAX_DIMA];
  int parity;
  int outFunc;                      // activation of the output layer (settings nnOutputFunc)
};

// recurrent (LSTM) hidden layers, rec.hip: per (sample, step) rows r = b * K + k, K = nnBPTTseq + 1
struct RecLayer {
  int nIn, nC, hasRes, resW;
  long long indW, indB, indWr, indBr;
  float* A; int ldA;       // [R][ldA]   [input of the step (nIn) | previous output (nC)]: A operand of the dW contraction
  float* X; float* Y;      // [R][4 nC]  cell input + gates / output, state, pre-gate cell output
  float* D;                // [R][4 nC]  deltas of cell input and gates: B operand of the dW contraction
  float* Rd; int ldR;      // [R][ldR]   delta at the residual output (hasRes)
  float* A2; int ldA2;     // [R][ldA2]  MGU: previous output x forget gate (A operand of the recurrent state weights)
};
struct RecArgs {
  DevScalars* sc; DevReplay rp; DevBatch bt;
  int B, dS, nL, K, nBPTT;
  int gates;                       // 4: LSTM (cell input, input / forget / output gate), 2: MGU (forget gate, state), 1: dense layer with a
                                   // recurrent term (nnType "RNN": BaseLayer with bRecurrent, Layer_Base.h:64-113)
  int func;                        // gates == 1: the layers' activation (settings nnFunc)
  const float* W;
  RecLayer L[HL_MAX_HIDDEN];
  float* Yout; int ldY;            // output of the last block at the sampled step (rows < B) and at t+1 (next rows): input of the head
  const float* Dres; int ldD;      // head: gradient w.r.t. Yout, rows < B
  const float* actStates; int actSteps;   // acting (hl_forward_sequence): raw states of the agent's last steps instead of a minibatch
  int actCtx;                      // ... of which the first actCtx lie in front of the window (they only feed appended observations)
  int nApp;                        // appended observations: the first layer's input is the step's state followed by the nApp before it
  const float* Xin; int ldXin;     // != nullptr: the first layer's input rows, written by launches in front (conv stack): row b K + k, next rows behind B K
  // a stack of two layer types runs as two launches (lower segment: the rnn kernels):
  // time-step-major LSTM (rectm.hip): window geometry per sample and the errors carried between the (layer, step) launches
  int* tmT; int* tmSteps; int* tmNext;                              // [B] steps in front of the sampled one / forward steps (next state included) / row of the next state's output
  float* tmER[HL_MAX_HIDDEN]; float* tmSD[HL_MAX_HIDDEN]; float* tmFP[HL_MAX_HIDDEN];      // [B][nC]: error handed back by step k + 1 / LSTM: state delta of step k + 1, MGU: dLdO of the step / MGU: W_sr dS of the step
  float* tmET[HL_MAX_HIDDEN];      // [B][nC]: backward by diagonals -- the error from the layer above at this step, left by whichever producer of a tile of deltas arrives first
  unsigned* tmCtr; int tmCtrN; int tmCtrOff[HL_MAX_HIDDEN];      // LSTM backward by diagonals: one arrival counter per (layer, 16-cell tile, 16-sample block), the layer's first one
  float* YoutRows; int ldYR;       // != nullptr: the last block's output of EVERY window step goes here (row b K + k, next rows behind B K): the upper segment's Xin
  const float* DresRows; int ldDR; // != nullptr: gradient w.r.t. those outputs per window row, from the upper segment (instead of Dres at the sampled step only)
};
struct WinRowsArgs { const DevScalars* sc; DevScalars* scW; const long long* slot; const int* t; const int* nextSrc; int B, K, nBPTT, parity;
                     long long* slotW; int* tW; int* nextSrcW; };
hipError_t launch_window_rows(const WinRowsArgs& a, hipStream_t s);
hipError_t launch_rec_forward(const RecArgs& a, hipStream_t s);
hipError_t launch_rec_backward(const RecArgs& a, hipStream_t s);
bool rec_tm_ok(const RecArgs& a);                                     // rectm.hip serves this net: a launch per (layer, window step) over the whole minibatch
bool rec_tm_act_ok(const RecArgs& a);                                 // ... and this acting window (layers beyond 256 cells)
hipError_t launch_rec_tm_forward(const RecArgs& a, hipStream_t s);
hipError_t launch_rec_tm_backward(const RecArgs& a, hipStream_t s);
// window forward + output layer + head + back-propagation through time of a sample as ONE launch (rec.hip: lstm32_step_wave_kernel)
struct ExtraArgs;
bool rec_step_fused_ok(const RecArgs& a, const HeadArgs& ha);
hipError_t launch_rec_step_fused(const RecArgs& a, const HeadArgs& ha, const ExtraArgs* extra, hipStream_t s);

// convolutional preprocessing (conv.hip): one layer's geometry and buffers
struct ConvGeo {
  int InC, InY, InX, KnC, KnY, KnX, S, OpY, OpX;
  int K, P;                  // InC KnY KnX (patch size), OpY OpX (output positions)
  int ldIn, ldOut;           // row pitch of the input / output activation arrays
  long long indW, indB;      // filter [KnC][InC][KnY][KnX] and bias [KnC][OpY][OpX] in the parameter blob
  const float* in;           // [rows][ldIn]  input images (first layer: the standardised stacked states)
  float* X; float* Y;        // [rows][ldOut] pre-activation / output, [c][oy][ox] per row
  float* D;                  // [B][ldOut]    dL/dX of this layer
  float* part;               // [nChunks][KnC K] partial filter gradients
  float* Wf; float* Wx;      // the filters in the kernels' LDS layouts (conv.hip: conv_prep_kernel)
  int nChunks, chunkRows, dwBlock0;   // reduction chunks of the filter gradient; first workgroup of this layer in the dW launch
  // row-block kernels (conv.hip: conv_fwd_rows_kernel / conv_dw_rows_kernel; layers with a large input image): a workgroup owns
  // rbRows output rows of one sample and stages the rbWin input rows under them in LDS.  rbRows = 0: not used for this layer.
  int rbRows, rbCount, rbWin;
  int rbKind;                // 1: the first layer of RACER_atari.json read from the replay -- kernels with the geometry at compile time (set per launch: convArgs)
  // filter gradient with the operands staged in LDS (conv.hip: convDwStaged; layers behind the first): a workgroup owns dwG rows
  // (samples) x one tile of 16 channels and leaves one partial per group of rows.  0: the gather workgroups (tile, chunk) above.
  int dwG;
};
// where the row-block kernels of the first layer take their input rows from when no stacked minibatch rows X0 were written
// (training steps: the replay itself, standardised on the way -- stack_gather_kernel's mapping, conv.hip)
struct ConvSource {
  int on;                       // 0: ConvGeo::in (X0 rows)
  const float* S; const float* mean; const float* scale; int dS, nApp;
  const long long* slot; const int* t; const int* nextSrc;
};
struct ConvArgs {
  DevScalars* sc; int parity, B, nL;
  const float* W; float* Wrw; float* M1; float* M2; float* G;
  ConvGeo L[HL_MAX_CONV];
  ConvSource src;
};
struct StackGatherArgs { DevScalars* sc; DevReplay rp; DevBatch bt; int B, dS, nApp, parity; float* X0; int ldX0; };
struct AdamHyper;
hipError_t launch_stack_gather(const StackGatherArgs& a, int maxRows, hipStream_t s);
hipError_t launch_extras_copy(const DevScalars* sc, int parity, const float* X0, int ldX0, int col0, int n, float* dst, int ldDst, int maxRows, hipStream_t s);
hipError_t launch_conv_prep(const ConvArgs& a, hipStream_t s);              // filters -> LDS layouts (once per step)
long long conv_prep_floats(const ConvGeo& g, int which);                   // floats of Wf (0) / Wx (1)
hipError_t launch_conv_forward(const ConvArgs& a, int l, int maxRows, hipStream_t s);
// weight-gradient tiles [tile0, tile1) of a dW problem table riding behind the workgroups of another launch (dw_wide_dev.h)
struct GemmProblem;
struct DenseRide { const GemmProblem* probs; int nProbs, tile0, tile1, own; AdamHyper hyp; };
hipError_t launch_conv_dx(const ConvArgs& a, int l, hipStream_t s, const DenseRide* ride = nullptr);       // D of layer l-1 from D of layer l
bool conv_dx_rides(const ConvGeo& g);
hipError_t launch_conv_dw(const ConvArgs& a, int totalBlocks, hipStream_t s);
int conv_row_block(const ConvGeo& g, int* win);
int conv_dw_staged_group(const ConvGeo& g, int B);                           // rows per workgroup of the LDS-staged filter gradient (0: gather form)
bool conv_rows_atari(const ConvGeo& g, const ConvSource& src);                // ... and their form with the geometry at compile time serves this layer / source
bool conv_rows_ok(const ConvGeo& g);                                          // the shape the row-block kernels are instantiated for                               // rows per workgroup of the row-block kernels (0: layer not served)
hipError_t launch_conv_forward_rows(const ConvArgs& a, int l, int maxRows, hipStream_t s);
hipError_t launch_conv_dw_all(const ConvArgs& a, int l, int dwBlocks, hipStream_t s);      // launch_conv_dw_rows(l) and launch_conv_dw as one launch
hipError_t launch_conv_dw_rows(const ConvArgs& a, int l, hipStream_t s);
hipError_t launch_conv_reduce_adam(const ConvArgs& a, const AdamHyper& hyp, int fuseAdam, hipStream_t s);

// convt.hip: the layers behind the first one by workgroups that keep a sample's maps and deltas in LDS
struct ConvTailPlan {
  int on;                       // 0: conv.hip's per-layer launches serve the stack
  int icPer[HL_MAX_CONV];       // input gradient of layer l: input channels per pass of the T buffer
  int bufD;                     // floats of one delta buffer (two of them: layer l reads one, writes the other)
  int ldsBack;                  // bytes of LDS of conv_back_kernel
  int atari;                    // the stack behind the first layer is RACER_atari.json's: kernels with the geometry at compile time
};
bool conv_tail_plan(const ConvGeo* L, int nL, ConvTailPlan* pl);
hipError_t launch_conv_back(const ConvArgs& a, const ConvTailPlan& pl, hipStream_t s);
hipError_t launch_conv_fwd_tail(const ConvArgs& a, const ConvTailPlan& pl, int maxRows, hipStream_t s);      // (pl.atari) layers 1 .. 3 forward, one launch      // D of layers nL-2 .. 0 from D of layer nL-1

struct PostArgs {
  DevScalars* sc; DevReplay rp; DevBatch bt;
  int B, mode;                       // mode bits: 1 aggregates, 2 beta+counters, 4 init (beta only)
  double clipImpWeight, epsAnneal, penalTol, maxObsGlobal, batchGlobal;
  int nRanks;
  int parity;                        // buffer of the step being closed; etaEff[parity^1] is written
  float eta0;
  int aggStaged;                     // 1: bt.aggIn holds the episode aggregates (fused kernel), no gather needed
  int hasAdv;                        // 1: Q = bt.newQ (head with an advantage), 0: Q = V
  int aggChunk;                      // large batches: 1 = the episode records of 256 samples per workgroup, nothing else; 2 = done by the launch in front (tail_dev.h: postPart)
  float* cntMsg;                     // != nullptr: the four replica counters travel inside the gradient message (16 floats, four
                                     // 16-bit chunks each: exact in fp32 for up to 256 replicas) instead of a collective of their own
};
enum { POST_AGG = 1, POST_BETA = 2, POST_INIT = 4, POST_ENCODE = 8 /* write the counters message only */,
       POST_DEFER = 16 /* with POST_AGG | POST_BETA: leave the far-policy count and the beta / alpha update that hangs off it to
                          farBetaPhase -- a rider of the NEXT step's fused kernel, whose heads wait for it (tail_dev.h) */ };

struct AdamArgs {
  const DevScalars* sc; float* W; float* M1; float* M2; const float* G; long long n;
  float eta0, lambda, fac; double epsAnneal; int parity;
};

// one-kernel sum over the replicas' windows (xchg.hip)
constexpr int XCHG_CHUNKS = 64;         // workgroups (= independently flagged chunks) of one collective, at most
constexpr int XCHG_CHUNKS_NODE = 32;    // ... as cut where every replica has a device of its own (hl_xchg_connect: fewer where they share one)
constexpr int XCHG_MAX_RANKS = 16;
struct XchgArgs {
  void* msg; long long n;                       // local message, summed in place
  int nRanks, rank;
  unsigned char* const* peers;                  // [nRanks] windows as this device addresses them (device array; [rank] = the own one)
  size_t slotsOffset, slotBytes;
  XchgCtl* ctl; DevScalars* sc;
  long long timeoutTicks;                       // wall_clock64 ticks (100 MHz) a workgroup waits for a peer's stamp
  int fuse; AdamArgs adam; PostArgs post;      // fuse != 0 (float messages): Adam on the summed chunk, then the bookkeeping pass
  long long pushed;                             // leading elements of the message the producing launch already stored into the peers' windows (PushArgs)
  int maxChunks;                                // workgroups (chunks) of a collective at most: XCHG_CHUNKS, fewer where replicas share a device (hl_xchg_connect)
};
int xchg_chunks(long long bytes, int maxChunks);      // how a message is cut: part of the wire protocol (every replica, folded or not, cuts alike)
hipError_t launch_xchg_allreduce(const XchgArgs& a, int dtype /* 0 float, 1 double, 2 int64 */, hipStream_t s);
// zeroes, in this replica's own window, what the collective just finished left in the senders' slots (first `bytes` of each): the
// gradient slots then hold zeros wherever no tile of a pushing launch writes (padding of the parameter layout)
hipError_t launch_xchg_clean(unsigned char* win, size_t slotsOffset, size_t slotBytes, int nRanks, const XchgCtl* ctl, long long bytes, hipStream_t s);
// extra workgroup appended to an MLP kernel's grid (tail_dev.h): role 0 none, 1 sampler phases
// (PH_A/B/C mask) of the NEXT step's minibatch, 2 bookkeeping of the step just computed
// PH_PUBLISH: phase C stops after the index -> (episode, step) search and hands the gather to helper
// workgroups (gatherHelper) through DevScalars::gatherFlag
enum { PH_A = 1, PH_B = 2, PH_C = 4, PH_ALL = 7, PH_PUBLISH = 8 };
struct ExtraArgs { int role; int phases; SampleArgs samp; PostArgs post; int helpers; /* head kernel: gather helper workgroups behind the rider (PH_PUBLISH) */ };
// conv_dw_all plus the dense layers' weight-gradient tiles (one workgroup per tile, dw_wide_dev.h) and that launch's rider as ONE launch
hipError_t launch_conv_dw_dense(const ConvArgs& a, int l, int dwBlocks, const GemmProblem* dProbs, int nProbs, int nTiles, const AdamHyper& hyp,
                                const ExtraArgs* extra, hipStream_t s);

// fused forward + head + dX kernel of the two-hidden-layer MLP (fused.hip)
struct FusedArgs {
  DevScalars* sc; DevReplay rp; DevBatch bt;
  int B, dS, dA, nDense, nOut, H, parity, func, resN;
  const float* X0; int ldX0;              // standardized states of the minibatch [Mmax][ldX0]
  const float* W;                         // weight blob
  long long indW0, indB0, indW1, indB1, indWr, indBr, indWo, indBo, indBp; int ldW0, ldW1;
  float* Y1; float* D1; float* Dres1; int ldA0;               // hidden block 0: activations, deltas
  float* X2; float* R2; float* D2; float* Dres2; int ldA1;    // hidden block 1: pre-activations (exchange), y3, deltas
  // a THIRD equal hidden block (fusedw.hip, nLH == 3: settings/RACER_glider.json): its weights, f'(x), block output, deltas
  int nLH; long long indW2, indB2, indWr2, indBr2; int ldW2, resN2;
  float* X3; float* R3; float* D3; float* Dres3; int ldA2;
  float* dOut; int ldDo;                  // output-layer deltas [B][ldDo]
  unsigned* panelCtr;                     // [panels][32] arrive counters of the panel barrier (monotonic)
  int variant;                            // development: stop after phase `variant` (0 = run everything)
  int deferBeta;                          // 1: beta of this step is still being computed by the rider in block 1 (POST_DEFER): the heads wait for DevScalars::betaSeq
  int xcdSafe;                            // 1: the workgroups of a panel may sit on different XCDs (probe at hl_create): the panel exchange
                                          // goes through agent-scope stores / loads instead of plain ones through the shared L2
  unsigned long long boundedMask;         // bit i: action component i is bounded (dA <= 7 here)
};

struct EpisodeSweepArgs {   // Retrace / updateCumulative over episodes
  DevScalars* sc; DevReplay rp;
  const int* eids; int count;        // eids == nullptr: all current positions (count = nEpisodes)
  float gamma, lambda; int recompute; // recompute=1: Episode::updateCumulative first
  int skipRetrace;                   // 1: aggregates only (restart from a checkpoint keeps the stored estimates)
  float* redMaxAbs;   // per-block partials (recompute only)
  int retKind;                       // HL_RET_*: computeRetrace / computeRetraceExplBonus / computeGAE (MemoryProcessing.cpp:391-416)
  double* redErr;                    // per-block sums of the squared changes of the estimates (recompute only: the dRet column)
};

struct MomentsArgs {
  DevScalars* sc; DevReplay rp; int dS; int nEpisodes;
  double* partial; int nBlocks;      // [nBlocks][2dS+3]
  double* moments;                   // [2dS+3]
  int bInit; double learnrate, epsAnneal, rRateFac;
};

// dense layers at large local batches (bigmm.hip): weight-stationary forward / dX panels, 64 x 64 weight-gradient tiles over row chunks
bool big_panel_ok(const GemmProblem& P);
hipError_t launch_big_panel(const GemmProblem& P, const DevScalars* sc, int parity, hipStream_t s);
bool big_mm_ok(const GemmProblem& P);
hipError_t launch_big_mm(const GemmProblem& P, const DevScalars* sc, int parity, hipStream_t s);
bool big_dw_ok(const GemmProblem& P);
int big_dw_chunk_rows(const GemmProblem& P);
int big_dw_blocks(const GemmProblem& P);
hipError_t launch_big_dw(const GemmProblem* dProbs, const BigDwList& L, hipStream_t s);      // (indices relative to dProbs)
// output layer + head + dX of the output layer for panels of 16 samples (headp.hip): the throughput form of launch_head
bool panel_head_ok(const HeadArgs& a);
hipError_t launch_panel_head(const HeadArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s);
hipError_t launch_sample(const SampleArgs& a, hipStream_t s);
hipError_t launch_post_agg_chunks(const PostArgs& a, hipStream_t s);
// one launch, two independent workgroups: bookkeeping of a finished step (post) and sampling of a
// minibatch (samp); either may be nullptr
hipError_t launch_step_tail(const PostArgs* post, const SampleArgs* samp, hipStream_t s, int phases = PH_ALL);
enum { GEMM_ROLE_FWD0 = 0, GEMM_ROLE_FWD = 1, GEMM_ROLE_DX = 2, GEMM_ROLE_DW = 3 };
constexpr int DW_TABLE_MAX = 8;
struct DwTable { GemmProblem p[DW_TABLE_MAX]; int n; };
// the replicas' exchange folded into the weight-gradient launch (round 6): nCh chunk workgroups behind the nTiles tile workgroups stamp,
// wait, sum in rank order, apply Adam and close the step (xchg_dev.h: xchgChunk<float, true, true>); windows and ranks: AdamHyper::push
struct FoldArgs { int on, nCh, nTiles, pad; float* msg; long long n; float* W; float* M1; float* M2; long long nAdam; XchgCtl* ctl; long long timeoutTicks; };
// riders: `extra` (bookkeeping of this step) in workgroup 0, `extra2` (sampler phase C of the next minibatch, PH_PUBLISH) behind it,
// followed by extra2->helpers gather workgroups
hipError_t launch_dw_table(const DwTable& tbl, int nBlocks, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s,
                           const ExtraArgs* extra2 = nullptr, const FoldArgs* fold = nullptr);
// up to two riders (extra, extra2) occupy workgroups 0 and 1 of the grid
hipError_t launch_gemm(int role, const GemmProblem* dProbs, int nProbs, int nBlocks, const DevScalars* sc,
                       const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s, const ExtraArgs* extra2 = nullptr);
// weight gradients over many rows (recurrent nets) as one launch of 16-wavefront workgroups, no split-row join (gemm16.hip: dw_wide_kernel)
constexpr int DW_WIDE_Q = 4;      // row quarters per tile: `part` holds nTiles (rounded up to 8) x DW_WIDE_Q x 256 floats, `ctr` one counter per tile (zeroed once)
hipError_t launch_dw_wide(const GemmProblem* dProbs, int nProbs, int nTiles, int nq, float* part, unsigned* ctr, const DevScalars* sc, const AdamHyper& hyp,
                          const ExtraArgs* extra, const ExtraArgs* extra2, hipStream_t s);      // nq: 1 (a tile's rows in one workgroup) or DW_WIDE_Q
hipError_t launch_splitk_reduce(const GemmProblem* dProbs, int nProbs, int maxMN, const DevScalars* sc, const AdamHyper& hyp, hipStream_t s,
                                const PostArgs* farBeta = nullptr);      // farBeta: a rider workgroup runs farBetaPhase (tail_dev.h)
hipError_t launch_head(const HeadArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s);
hipError_t launch_fused(const FusedArgs& a, int maxRows, const ExtraArgs* extra, hipStream_t s);
size_t fused_lds_bytes(int dS, int H);
// the same step for two-hidden-layer nets with wide states (first layer streamed in slabs) and any head of head_rows.h (fusedw.hip)
hipError_t launch_fused_wide(const FusedArgs& a, const HeadArgs& ha, int maxRows, const ExtraArgs* extra, hipStream_t s);
bool fused_wide_ok(int dS, int H, int nDense, int nOut, int ldWo, int nAdv, int comps, int nLH = 2);      // nLH: two or three equal hidden blocks
size_t fused_wide_lds_bytes(int dS, int H, int nDense, int nOut, int ldWo, int nAdv, int nLH = 2);
int fused_wide_threads();
int fused_threads();
hipError_t launch_post(const PostArgs& a, hipStream_t s);
hipError_t launch_empty(hipStream_t s);
// episode ingestion (misc.hip: ingest_kernel): a batch of finished episodes staged in pinned host memory -- descriptor table
// first, then per episode states f32 | actions f64 | behaviour policies f64 | rewards f64 | values f32 | advantages f32, each
// block 16-byte aligned -- is scattered into the replay's structure of arrays by ONE kernel that reads the host buffer directly
struct IngestDesc { long long off, tag; unsigned long long data; int N, eid, term; float totR; };   // data: byte offset of the episode's block
struct IngestArgs {
  DevReplay rp; const unsigned char* stage; int nEp, dS, dA, polDim;
  const double* stats; int nEpTable;     // stats[1] = avgSquaredErr over the nEpTable episodes of the device table (placeholder error of new episodes)
};
constexpr int INGEST_MAX_EP = 1024;
hipError_t launch_ingest(const IngestArgs& a, hipStream_t s);
// forward / dX tile with the whole reduction in flight at once (gemm16.hip: gemm_os_kernel), 256 < K <= 640
bool gemm_oneshot_ok(int flavor, int K);
// all dense forward layers in one launch (gemm16.hip: fwd_chain_kernel)
size_t fwd_chain_lds_bytes();
int fwd_chain_blocks(int maxRows, int HT);
hipError_t launch_fwd_chain(const GemmProblem* dProbs, const int* idx, int nLayers, int HT, int maxRows, unsigned* panelCtr, const DevScalars* sc,
                            const AdamHyper& hyp, const ExtraArgs* extra, const ExtraArgs* extra2, hipStream_t s);
// forward chain + head + input-gradient chain in one launch (gemm16.hip: step_chain_kernel); xIdx: the dX problems from the last hidden layer down
hipError_t launch_step_chain(const GemmProblem* dProbs, const int* fIdx, int nF, const int* xIdx, int nX, int HT, int maxRows, unsigned* panelCtr,
                             const HeadArgs& ha, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s);
hipError_t launch_gemm_oneshot(int role, const GemmProblem* dProb, int K, int nBlocks, const DevScalars* sc, const AdamHyper& hyp, const ExtraArgs* extra, hipStream_t s);
struct TouchArgs { const void* ptr[24]; long long bytes[24]; int stride[24]; int n; float* sink; };
hipError_t launch_touch(const TouchArgs& a, hipStream_t s);      // reads one word per 4 KB of each array (address translations resident)
hipError_t launch_set_ret_counters(DevScalars* sc, long long cnt, hipStream_t s);   // the statistics line consumed the return-estimate counters (MemoryBuffer.cpp:534-544)
// XCD of every workgroup of a launch shaped like the fused kernel's (fused.hip: xcc_probe_kernel): out[block] = HW_REG_XCC_ID
hipError_t launch_xcc_probe(int nBlocks, int nThreads, size_t ldsBytes, int* out, hipStream_t s);
hipError_t launch_notify(DevScalars* sc, unsigned* hostWord, hipStream_t s);   // ++sc->notifySeq -> pinned host word (completion stamp polled by hl_sync)
hipError_t launch_rng_restore(DevScalars* sc, hipStream_t s);   // DevScalars::rngBak -> rng (a pre-sampled minibatch is discarded)
hipError_t launch_act_standardize(DevScalars* sc, DevReplay rp, const float* S, int n, int dS, int dIn, float* X0, int ldX0, hipStream_t s);
hipError_t launch_act_output(const float* Y, int ldY, int H, const float* W, long long indWo, long long indBo, long long indBp, int ldWo,
                             int nDense, int dA, int n, double* O, hipStream_t s, unsigned* done = nullptr, unsigned tag = 0, int outFunc = 0);
hipError_t launch_adam(const AdamArgs& a, hipStream_t s);
// rollout inference for a few agents (misc.hip: act_forward_kernel): the whole dense network for one raw state per workgroup, states
// read from and outputs written to pinned host memory, completion stamped per row
constexpr int ACT_MAXW = 1024, ACT_MAXROWS = 64;
struct ActLayer { int nIn, size, ldW, func, hasRes, resW; long long indW, indB, indWr, indBr; };
struct ActArgs {
  const float* W; const float* stMean; const float* stScale;
  const float* in; double* out; volatile unsigned* done; unsigned tag;      // pinned host memory (device-mapped)
  int dS, dIn, nL, nDense, nSig, nOut, ldWo; long long indWo, indBo, indBp;
  int outFunc;
  ActLayer L[HL_MAX_HIDDEN];
};
hipError_t launch_act_forward(const ActArgs& a, int n, hipStream_t s);
hipError_t launch_episode_sweep(const EpisodeSweepArgs& a, int nBlocks, hipStream_t s);
hipError_t launch_far_build(DevReplay rp, int nEpisodes, hipStream_t s);    // after the table or all fractions changed
hipError_t launch_sweep_finish(DevScalars* sc, DevReplay rp, const float* redMaxAbs, const double* redErr, int countRet, int nBlocks, hipStream_t s);
hipError_t launch_moments(const MomentsArgs& a, hipStream_t s);          // partial sums + final sum
hipError_t launch_moments_apply(const MomentsArgs& a, hipStream_t s);    // EMA update of the scaling
hipError_t launch_set_counts(DevScalars* sc, long long nTransitions, long long nEpisodes,
                             long long seenEps, long long seenSteps, hipStream_t s);
hipError_t launch_episode_max(DevScalars* sc, DevReplay rp, int nEp, hipStream_t s);   // sc->maxAbsErrAll = max over the stored episodes
hipError_t launch_stats(DevScalars* sc, DevReplay rp, int nEpisodes, double* out /*16 doubles*/, hipStream_t s);
// prioritised samplers (per.hip): probabilities, ranking and the sequential normalisation / cumulative table
struct PerArgs {
  DevReplay rp; int nEpisodes, algo;
  float* prob; double* cp;                                   // [capSlots] (PERseq: [episodes])
  float *key, *keySorted; unsigned *idx, *idxSorted;         // PERrank: squared errors and flat indices, sorted descending (stable)
  void* temp; size_t tempBytes;
  void* scan; size_t scanBytes;                              // scratch of the table's grid form (per_scan_scratch_bytes)
};
hipError_t launch_per_prepare(const PerArgs& a, long long nTransitions, hipStream_t s);
size_t per_sort_temp_bytes(long long n);
size_t per_scan_scratch_bytes(long long n);
hipError_t launch_per_scan(const float* prob, double* cp, long long n, int which, void* scratch, hipStream_t s);      // (tests: the table of any probability array; which = 0 grid form where long enough, 1 walk, 2 one workgroup)
struct HistArgs { DevReplay rp; int nEpisodes; float bounds[82]; unsigned long long* counts; };
hipError_t launch_impw_hist(const HistArgs& a, hipStream_t s);
int sweep_blocks(int count);
int moments_blocks(int nEpisodes);

}  // namespace hl
