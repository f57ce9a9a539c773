// smarties_amd/csrc/hl_types.h -- structures shared by the host learner (learner.cpp) and the
// gfx950 kernels (kernels.hip).  Device-resident state of one learner replica.
#pragma once
#include <stdint.h>
#include "../../include/smarties_hip.h"

namespace hl {

// ---------------------------------------------------------------------------
// Device-resident scalars of the learner (reference: MemoryBuffer.h:41-44 beta/Cmax/Cinv,
// ReplayStats / ReplayCounters (ReplayStatsCounters.h), Optimizer.h:96 beta_t, mt19937
// generators[0] (ExecutionInfo.cpp:391)).  Only single-block kernels write it.
// ---------------------------------------------------------------------------
struct DevScalars {
  double beta, alpha, Cmax, Cinv;
  double adam_bt1, adam_bt2;
  double maxAbsErrEMA;            // ReplayStats::maxAbsError
  long long nStep;                // AdamOptimizer::nStep (completed prepare_update calls)
  long long nGradSteps;
  long long nFarTotal;            // ReplayStats::nFarPolicySteps of this replica as of the last statistics pass (dev_common.h: farExact)
  long long nFarStat;             // ReplayStats::nFarPolicySteps: value of the last statistics pass
  long long nTransitions;         // ReplayCounters::nTransitions (this replica)
  long long nEpisodes;
  long long cnt[4];               // {seenEps, seenSteps, nFar, nStored}: local, then all-reduced (C2)
  long long seenLocal[2];         // this replica's seenEps / seenSteps (cnt[0..1] are reset from it each step)
  float rewMean, rewScale, rewStd;
  float maxAbsErrAll;             // max over episodes of Episode::maxAbsError (running; recomputed after arrivals / removals and by the sweeps)
  float maxAbsErrStep;            // ... as the statistics pass of the current step saw it, i.e. before that step's removals: what the
                                  // ReplayStats::maxAbsError average of the step takes in (MemoryProcessing.cpp:223,241)
  // the minibatch workspace is double buffered (sampling of step k+1 overlaps the update of step k)
  unsigned maxAbsScratch;         // large batches: max |error| of the step's episode records, collected by post_agg_chunks_kernel (float bits)
  int nNext[2];                   // rows B..B+nNext-1 of the minibatch hold truncated next states
  int nRows[2];                   // B + nNext
  float etaEff[2];                // Adam step size incl. bias correction for the step using buffer p
  int errFlag;                    // sticky device-side error code (0 = ok)
  long long gatherFlag[2];        // sampler -> gather-helper hand-off per minibatch buffer (value: nStep + 1)
  unsigned rngPos;
  unsigned rng[624];
  // generator state before the draws of the LAST pre-sampled minibatch (the sampler of step k+1 rides along step k;
  // if the host has to discard that minibatch -- new episodes, an eviction, explicit indices -- it puts this state back)
  unsigned rngBakPos;
  unsigned rngBak[624];
  long long seenUpd[2];           // seen episodes / steps (summed over the replicas) as of the last updateCounters: ReplayCounters::nSeenEpisodes,
                                  // nSeenTransitions (MemoryProcessing.cpp:60-61) -- what the stats line prints
  long long sampleSeq;            // minibatches drawn so far (sampler phase A): hand-off tag when the gather rides along the dW kernel
  // ReplayStats::sumReturnsEstimateErrors / countReturnsEstimateUpdates (MemoryProcessing.cpp:250-258): squared changes of the
  // return estimates in the 1000-step sweeps since the statistics line last printed them (hl_metrics resets; -1 = printed)
  double sumRetErr; long long cntRetUpd;
  long long betaSeq;              // nGradSteps for which farBetaPhase has published beta / alpha (POST_DEFER)
  unsigned notifySeq;             // exact-size graphs replayed so far (their last node stores it into pinned host memory: hl_sync)
  long long dbgT[32];             // development: wall_clock64() stamps of the tail phases
  long long dbgStep[128];         // development (-DHL_STEP_STAMPS): entry stamps of the two step kernels, by step number mod 64
};

// ---------------------------------------------------------------------------
// Replay buffer in HBM: structure-of-arrays over "slots" (one slot per stored state,
// Episode.h:66-82), episodes occupy contiguous slot ranges.
// ---------------------------------------------------------------------------
// everything the sampler needs about the episode at one position of the current order: the
// flat-index prefix (Sampling.cpp:26-47), slot offset, length, terminal flag, storage id and tag
struct PosRec {
  long long prefix;        // transitions stored before this episode
  long long off;           // first slot
  long long tag;
  int N;                   // number of states
  int eidTerm;             // eid | (terminated << 31)
};

// far-policy count (dev_common.h): episodes a thread of the bookkeeping rider keeps in registers, pieces its segment is walked in
constexpr int FAR_REGS = 24, FAR_SUB = 4, FAR_Q = FAR_REGS / FAR_SUB;

struct DevReplay {
  float* S;        // [cap][dS]      states (raw; standardized on gather, Episode.h:172-183)
  double* A;       // [cap][dA]      actions (f64 as in the reference, Episode.h:73)
  double* MU;      // [cap][2dA]     behaviour policy mean|stdev
  double* R;       // [cap]          rewards
  float *V, *ADV, *RET;    // stateValue, actionAdvantage, returnEstimator
  float *DQ, *IMPW, *DKL;  // deltaValue, offPolicImpW, KullbLeibDiv
  // per episode storage id (eid)
  long long* epOff;        // first slot
  int* epN;                // number of states
  unsigned char* epTerm;   // bReachedTermState
  long long* epTag;        // caller-supplied episode tag
  float* epAgg;            // [nEpCap][AGG_N] running aggregates (Episode.h:99-103)
  // current episode order (position -> eid) and transition prefix (Sampling.cpp:26-47)
  int* posEid;             // [nEp]
  long long* posPrefix;    // [nEp+1]
  struct PosRec* posRec;   // [nEp+1] the same table as one 32-byte record per position (sampler)
  float* stMean; float* stScale; float* stStd;   // [dS]
  // the far-policy count (dev_common.h): fracFarPolSteps and nsteps per table position in the layout of the walk
  // [nEpCap + 256 each], and the 256 segment starts of the last pass (first guess of the next one)
  float* farP; float* farN; unsigned long long* farStart;
};
enum { AGG_TOTR = 0, AGG_AVGKL, AGG_FRACFAR, AGG_AVGSQERR, AGG_MAXABSERR, AGG_SUMQ2, AGG_SUMQ,
       AGG_MAXQ, AGG_MINQ, AGG_USED, AGG_LEN = 9 /* staging only: episode length */, AGG_N = 12 };

// ---------------------------------------------------------------------------
// Minibatch workspace (MiniBatch.h) + taps
// ---------------------------------------------------------------------------
struct DevBatch {
  long long* flat;     // [B] sorted unique flat indices
  int* pos;            // [B] episode position
  int* eid;            // [B]
  int* t;              // [B] step within episode
  long long* slot;     // [B] replay slot of the sampled state
  int* nextOf;         // [B] row index (>= B) holding s_{t+1} if truncated, else -1
  int* nextSrc;        // [B] for next row j: sample b it belongs to
  long long* tag;      // [B] tag of the sampled episode
  unsigned* sVals;     // [B] candidate flat indices between the sampler phases (tail_dev.h)
  int *pEid, *pNextOf; // [B] copies made by the head kernel for the bookkeeping pass (which runs
                       //     concurrently with the sampling of the NEXT minibatch)
  // head outputs / write-back staging (old values are needed by the aggregate updates)
  double* O;           // [2B][nOut]
  double* G;           // [B][nOut]
  double *rho, *dkl, *dq;        // [B]
  unsigned char* far;            // [B]
  float *newDQ, *newDKL, *newW, *newV;   // [B] values written to the replay (Fval casts)
  float* newQ;                           // [B] Q = V + A of the sampled step (heads with an advantage; VRACER: Q = V)
  float *oldDQ, *oldDKL, *oldW, *oldV, *oldADV;   // [B]
  float *nextV, *oldNextV, *oldNextADV;           // [B] (indexed by sample b)
  float* gParam;       // [B][dA] gradient wrt the ParamLayer outputs
  float* aggIn;        // [B][AGG_N] aggregates (+ length in slot AGG_LEN) of the sampled episode as of the
                       //     start of the step, staged by the fused kernel for the bookkeeping pass
};

// one dense hidden block of the MLP (BaseLayer [+ ParametricResidualLayer])
struct DevHidden {
  int nIn, size, ldW;            // ldW = nOut_simd (Layer_Base.h:46)
  int func;
  long long indW, indB;          // dense weights / bias offsets in the blob
  int hasRes, resW;              // resW = min(nIn, size) (Layers.h:357)
  int lstm;                      // 1: LSTM layer of `size` cells (Layer_LSTM.h): W is [(nIn + size)][4 size], ldW = 4 size, bias 4 size
  long long indWr, indBr;        // residual w / b offsets
  float *X, *Y, *Rr;             // pre-activation, activation, residual output  [Mmax][ldA]
  float *D, *Dres;               // delta after act', gradient wrt block output  [B][ldA]
  int ldA;
};

// ---------------------------------------------------------------------------
// generic small-GEMM problem descriptor (see kernels.hip: gemm16_kernel)
// ---------------------------------------------------------------------------
enum { GEMM_F = 0,   // C[M,N] = A[M,K] * B[K,N]         A rows, B = weights [K][ldb]
       GEMM_X = 1,   // C[M,N] = A[M,K] * B^T,           B = weights [N][ldb] (reduce over its columns)
       GEMM_W = 2,   // C[M,N] = A^T * B,                A = acts [K][lda] (+ ones row M-1), B = deltas [K][ldb]
       RED_COL = 3 };// out[j] = sum_m A[m][j] * (B ? B[m][j] : 1)
enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2, EPI_NONE = 3 };

struct GemmProblem {
  int flavor, epi;
  int M, N, K;           // output M x N, reduction length K
  int dynRows;           // 1: number of valid rows of A (and C) = DevScalars::nRows (GEMM_F only)
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;     // primary output
  // epilogue operands
  const float* bias;     // EPI_FWD
  float* C2; float* C3;  // EPI_FWD: Y, R ; EPI_DX: D (C = Dres)
  const float* resW; const float* resB; const float* resIn; int ldRes; int resN;  // residual
  const float* actX; const float* actY; int ldAct; int func;                        // EPI_DX
  float* biasOut;        // EPI_DW: row M-1 of the product (the ones row) goes here
  // fused Adam (single replica): parameter / moment arrays laid out like C and like biasOut
  int adam;
  float *adW, *adM1, *adM2;
  float *adbW, *adbM1, *adbM2;
  int tileStart, tilesM, tilesN;
  // GEMM_W over many rows (recurrent nets: batch x BPTT steps): the reduction is cut into nSplit chunks of 256 rows, one
  // workgroup per (tile, chunk); partial tiles go to part[chunk][M][N] and are summed in chunk order -- with the Adam
  // update, adamRed -- by splitk_reduce_kernel (gemm16.hip)
  int nSplit, adamRed;
  float* part;
  int bigChunk;          // large batches (bigmm.hip: big_dw_kernel): rows per chunk; 0: the problem is served by the common launch
};

// the problems of the large-batch weight-gradient launch (bigmm.hip: big_dw_kernel): indices into the problem table, first workgroup of each
constexpr int BIG_DW_MAX = 12;
struct BigDwList { int n; int idx[BIG_DW_MAX]; int start[BIG_DW_MAX + 1]; };

// one-kernel exchange between replicas (xchg.hip): sequence number of the next collective, arrival count of its workgroups
// (a cache line per role: the sequence number is read by every pushing tile, `pushed` takes an atomic from each of them, `ready` is
//  polled by the chunk workgroups -- on one line the pollers and the tiles' atomics queued behind each other: 62 against 38 us per step)
struct alignas(128) XchgCtl {
  unsigned long long seq; unsigned long long pad0[15];
  unsigned int done; unsigned int arrived; /* workgroups of the running collective whose peers' stamps all came (FUSE: nobody applies Adam before all have) */ unsigned int pad1[30];
  unsigned int pushed; /* folded weight-gradient launch: its producers (tiles, bookkeeping rider) whose window stores are acknowledged */ unsigned int pad2[31];
  unsigned long long ready; /* = seq + 1 once the last producer of the folded launch of collective `seq` has arrived */ unsigned long long pad3[15];
};
// replicas connected through peer windows: the weight-gradient launch stores every gradient tile into the peers' windows as well
// (16-byte stores over xGMI from the tile's epilogue), so the transfer overlaps the launch and the exchange kernel behind it only
// stamps, waits, sums and applies Adam (round 4; before: the exchange kernel pushed the whole message after the launch)
struct PushArgs {
  int on, nRanks, rank, self;           // self: the own window takes the values too (folded launch: the chunk workgroups read windows only)
  unsigned char* const* peers;          // [nRanks] windows as this device addresses them
  unsigned long long slotsOffset, slotBytes;
  const XchgCtl* ctl;                   // parity of the collective the gradient belongs to = ctl->seq & 1
  const float* gBase;                   // gradient array (offsets inside the message)
};
struct AdamHyper { float eta0, lambda, fac; double epsAnneal; int parity; /* minibatch buffer of this step */
                   int variant; /* development ablation switches, 0 in production */
                   PushArgs push; };

}  // namespace hl
