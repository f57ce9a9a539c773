/*
 * include/smarties_hip.h -- C-ABI of libsmarties_hip.so, the MI355X-native
 * replacement for the learner-update hot path of cselab/smarties
 * (V-RACER / ReF-ER off-policy update over a device-resident replay buffer).
 *
 * Plain C: pointers, sizes and PODs only.  No torch / HIP types cross this
 * boundary.  Every entry point returns an int status (HL_OK == 0); nothing in
 * the library aborts the process -- a C++ RACER-shaped host class converts a
 * non-zero status into smarties' die() (reference: Utils/Warnings.h:34-44).
 *
 * What each group of entry points replaces in the reference
 * (paths relative to /root/reference/source/smarties/):
 *
 *   hl_create / hl_destroy        RACER ctor + setupNet + Builder::build
 *                                 (Learners/RACER_common.cpp:71-115,
 *                                 Network/Builder.cpp:119-170,
 *                                 Network/Approximator.cpp:179-229) and the
 *                                 MemoryBuffer ctor (ReplayMemory/MemoryBuffer.cpp:23-46)
 *   hl_init_weights               Layer::initialize draw order
 *                                 (Network/Layers/Layer_Base.h:115-141)
 *   hl_set/get_params             Parameters blob, padded layout
 *                                 (Network/Layers/Parameters.h:159-176)
 *   hl_append_episode             MemoryBuffer::addEpisodeToTrainingSet +
 *                                 pushBackEpisode (MemoryBuffer.cpp:131-170,479-520)
 *   hl_initialize                 Learner::initializeLearner (Learners/Learner.cpp:47-72)
 *   hl_step / _begin / _end       RACER::setupTasks stepMain + stepComplete
 *                                 (Learners/RACER.cpp:81-108):
 *                                 Learner_approximator::spawnTrainTasks
 *                                 (Learner_approximator.cpp:36-92) ->
 *                                 MemoryBuffer::sampleMinibatch (MemoryBuffer.cpp:359-432),
 *                                 RACER::Train (RACER_train.cpp:14-67),
 *                                 Approximator::forward/backProp (Approximator.h:118-297),
 *                                 AdamOptimizer::prepare_update/apply_update
 *                                 (Network/Optimizer.cpp:110-178),
 *                                 Learner::processMemoryBuffer (Learner.cpp:74-100)
 *   hl_comm_*                     the MPI_Iallreduce calls C1-C3 of SURVEY.md 2.4
 *                                 (Network/Optimizer.cpp:116, Utils/DelayedReductor.cpp:75,80)
 *   hl_readback / hl_get_*        debug taps used by the parity tests
 *
 * The CPU oracle (oracle/port, test infrastructure only) exports the same
 * functions with the prefix ol_ and the same structs, so a parity test is the
 * same call sequence against two libraries.
 */
#ifndef SMARTIES_HIP_H
#define SMARTIES_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HL_API __attribute__((visibility("default")))

#define HL_MAX_DIMA   64
#define HL_MAX_HIDDEN 8
#define HL_MAX_CONV   8
#define HL_MAX_RANKS  256   /* replica counters travel as 16-bit chunks inside the fp32 gradient all-reduce */

/* status codes */
enum {
  HL_OK = 0,
  HL_ERR_BAD_ARG = 1,        /* null pointer, size mismatch, unsupported option      */
  HL_ERR_NO_DEVICE = 2,      /* no HIP device / extension not usable                  */
  HL_ERR_HIP = 3,            /* a HIP runtime call failed (see hl_last_error)         */
  HL_ERR_STATE = 4,          /* call sequence violated (e.g. step before initialize)  */
  HL_ERR_TOO_FEW_DATA = 5,   /* "Parameter minTotObsNum is too low" (Learner_approximator.cpp:38-41) */
  HL_ERR_COMM = 6,           /* RCCL failure                                           */
  HL_ERR_IO = 7,
  HL_ERR_UNSUPPORTED = 8
};

/* nnFunc names accepted by the reference (Network/Layers/Functions.h:643-668) */
enum { HL_FUNC_LINEAR = 0, HL_FUNC_TANH = 1, HL_FUNC_SOFTSIGN = 2, HL_FUNC_RELU = 3,
       HL_FUNC_LRELU = 4, HL_FUNC_SIGM = 5, HL_FUNC_HARDSIGN = 6, HL_FUNC_SOFTPLUS = 7,
       HL_FUNC_EXPPLUS = 8, HL_FUNC_EXP = 9 };

/* advantage head: which RACER instantiation (Learners/RACER.cpp:114-116) */
/* hidden layer type (Network/Builder.cpp:48-117): dense, or LSTM (Network/Layers/Layer_LSTM.h; BASELINE config 4).
 * HL_NN_LSTM: rec.hip (a workgroup -- two wavefronts for the shipped 2 x 32 shape -- per sample walks the BPTT window; cells <= 256 per layer). */
enum { HL_NN_FFNN = 0, HL_NN_LSTM = 1, HL_NN_MGU = 2 /* Layer_GRU.h: what a partially observable MDP gets when nnType is left FFNN (Approximator.cpp:221-223) */,
       HL_NN_RNN = 3 /* "RNN" / "Recurrent" (Builder.cpp:76-81): dense layers with a recurrent term, y_t = f(W x_t + W_rec y_{t-1} + b)
                        (BaseLayer with bRecurrent, Layer_Base.h:64-113) */ };

/* settings key returnsEstimator (Settings/HyperParameters.cpp:135; MemoryProcessing::createReturnEstimator,
 * ReplayMemory/MemoryProcessing.cpp:391-450): how the per-step return estimates are swept backwards over an episode */
enum { HL_RET_RETRACE = 0,          /* "retrace" (what "default" means for RACER / VRACER, AlgoFactory.cpp:134-135): computeRetrace (:391-400) */
       HL_RET_RETRACE_EXPLORE = 1,  /* "retraceExplore": computeRetraceExplBonus (:402-408), bonus (1 - gamma)(|Q_ret - Q| - maxAbsError) */
       HL_RET_GAE = 2,              /* "GAE": computeGAE (:410-416) */
       HL_RET_NONE = 3 };           /* "none": the estimates stay as stored (computeReturnEstimator returns at once, :455) */

/* advantage head (Learners/AlgoFactory.cpp:109-152): Math/Zero_advantage.h (VRACER), Math/Gaus_advantage.h (RACER,
 * continuous actions: network outputs [V | coef, L+ x dA, L- x dA | mean x dA | sigma parameter x dA]);
 * Math/Discrete_advantage.h + Math/Discrete_policy.h (RACER, discrete actions: outputs [V | A x nOptions |
 * policy logits x nOptions], no sigma layer; per-step policy vectors have nOptions entries instead of 2 dimA) */
enum { HL_ADV_ZERO = 0 /* VRACER */, HL_ADV_GAUSSIAN = 1 /* RACER continuous */,
       HL_ADV_DISCRETE = 2 /* RACER discrete */ };

/* which episodes leave an over-full replay: settings key ERoldSeqFilter (MemoryProcessing::getERfilterAlgo,
 * ReplayMemory/MemoryProcessing.cpp:261-298) */
enum { HL_ER_OLDEST = 0,       /* "oldest" / "default": first in, first out                                   */
       HL_ER_FARPOLFRAC = 1,   /* "farpolfrac": the episode with the largest fraction of far-policy steps      */
       HL_ER_MAXKLDIV = 2,     /* "maxkldiv":   the episode with the largest average D_KL                      */
       HL_ER_MINERROR = 3 };   /* "minerror":   the episode with the smallest average squared TD error         */

/* minibatch sampler: settings key dataSamplingAlgo (Sampling::prepareSampler, ReplayMemory/Sampling.cpp:298-340) */
enum { HL_SAMPLE_UNIFORM = 0,  /* "uniform":  Sample_uniform (:50-97)                                                     */
       HL_SAMPLE_PERRANK = 1,  /* "PERrank":  TSample_impRank (:101-170): probability 1 / sqrt(sqrt(rank of the squared TD error)) */
       HL_SAMPLE_PERERR = 2,   /* "PERerr":   TSample_impErr (:173-230): probability (delta^2 + eps)^(1/4)                  */
       HL_SAMPLE_PERSEQ = 3 }; /* "PERseq":   Sample_impSeq (:234-296): episodes by (avg squared error + eps)^(1/4) x length, step uniform */

/* episode ordering used for the flat-index -> (episode, step) prefix walk */
enum { HL_ORDER_STABLE = 0,     /* stable sort by ID, newest first (product semantics)  */
       HL_ORDER_REFERENCE = 1 };/* std::sort each step exactly as MemoryProcessing.cpp:336
                                   (oracle only: reproduces the reference's permutation) */

/* Convolutional preprocessing layer as the environment declares it (Conv2D_Descriptor, Core/StateAction.h;
 * Communicator::setPreprocessingConv2d, Communicator.cpp:136-162): image [inpFeatures][inpY][inpX] -> SoftSign
 * convolution -> [outFeatures][outY][outX], filter [outFeatures][inpFeatures][filtery][filterx], one bias per OUTPUT
 * ELEMENT (Network/Layers/Layer_Conv2D.h:37-40).  The reference instantiates seven shapes (Network/Builder.cpp:189-203);
 * the library takes any stride / filter with zero padding. */
typedef struct hl_conv2d {
  int32_t inpFeatures, inpY, inpX, outFeatures, outY, outX, filterx, filtery, stridex, stridey, paddinx, paddiny;
} hl_conv2d;

/*
 * Learner configuration = the settings/<name>.json Learner surface
 * (Settings/HyperParameters.cpp:132-171) + the MDP descriptor fields the hot
 * path reads (Core/StateAction.h:57-112) + process layout.
 */
typedef struct hl_config {
  uint32_t struct_size;              /* = sizeof(hl_config), checked                      */
  int32_t dimS;                      /* MDP.dimStateObserved                               */
  int32_t dimA;                      /* MDP.dimAction (continuous)                         */
  uint8_t bounded[HL_MAX_DIMA];      /* MDP.bActionSpaceBounded[i]                         */
  int32_t n_hidden;                  /* len(nnLayerSizes)                                  */
  int32_t hidden[HL_MAX_HIDDEN];     /* nnLayerSizes                                       */
  int32_t nnFunc;                    /* HL_FUNC_*                                          */
  int32_t adv_kind;                  /* HL_ADV_*                                           */
  int32_t batchSize;                 /* GLOBAL batch (split over ranks, HyperParameters.cpp:186-189); the local share <= 16384 */
  int64_t maxTotObsNum;              /* GLOBAL replay size (split over ranks, :196-197)    */
  int64_t minTotObsNum;              /* GLOBAL; 0 = maxTotObsNum (HyperParameters.cpp:191)  */
  double gamma, lambda;              /* Retrace                                            */
  double clipImpWeight;              /* ReF-ER C0                                          */
  double penalTol;                   /* ReF-ER D                                           */
  double epsAnneal;
  double learnrate;
  double nnLambda;                   /* AdamW decay                                        */
  double explNoise;
  double outWeightsPrefac;
  uint64_t randSeed;                 /* ExecutionInfo::randSeed (rank is added, ExecutionInfo.cpp:387) */
  int32_t n_ranks, rank;             /* learner replicas on this node                      */
  int32_t device_id;                 /* HIP device ordinal; -1 = rank % device count       */
  int32_t episode_order;             /* HL_ORDER_*                                         */
  int32_t ref_threads;               /* OMP threads T of the reference run being mirrored (default 1): it seeds T - 1 further
                                        generators from the main one (ExecutionInfo.cpp:392-393), which shifts the stream all
                                        samples and weights are drawn from by T - 1 draws; per Adam step the main generator
                                        gives one draw whatever T is (thread 0's, Network/Optimizer.cpp:139) */
  int32_t n_options;                 /* HL_ADV_DISCRETE: number of action options (MDP.maxActionLabel, 2..64) of the ONE
                                        discrete action variable (dimA = 1; actions hold label + 0.1 as in
                                        Core/StateAction.h:322-341, policies the nOptions probabilities); else 0 */
  int32_t nn_type;                   /* HL_NN_*: settings nnType of the hidden layers                  */
  int32_t nnBPTTseq;                 /* recurrent nets: steps of truncated BPTT (0 = the default, 16)   */
  int32_t nAppendedObs;              /* MDP.nAppendedObs (Communicator::setNumAppendedPastObservations): the network input is the
                                        observed state of step t followed by those of t-1 .. t-nAppendedObs
                                        (Episode::standardizedState, Episode.h:172-183; steps before the first: the first) */
  int32_t n_conv;                    /* MDP.conv2dDescriptors: convolutional layers ahead of nnLayerSizes
                                        (Approximator::buildPreprocessing, Approximator.cpp:231-271; BASELINE config 5) */
  hl_conv2d conv[HL_MAX_CONV];
  int32_t ERoldSeqFilter;            /* HL_ER_*: removal rule of an over-full replay (equal keys: the older episode goes;
                                        the reference's non-stable std::sort leaves that to the library implementation)  */
  int32_t dataSamplingAlgo;          /* HL_SAMPLE_*.  The prioritised samplers rebuild a std::discrete_distribution over all
                                        stored transitions (episodes) before every minibatch, as the reference does: its
                                        normalisation and cumulative table are sequential double-precision passes, kept
                                        sequential on the device so that the drawn indices are the reference's (PERerr, PERseq;
                                        PERrank ranks equal errors in storage order where the reference's non-stable sort leaves it open) */
  int32_t returnsEstimator;          /* HL_RET_*                                                                               */
  int32_t nnOutputFunc;              /* HL_FUNC_*: activation of the output layer, settings key nnOutputFunc (Approximator.cpp:193,228;
                                        default "Linear").  Its inverse also shapes the initial output biases (Layer_Base.h:122-125) */
  int32_t n_encoder;                 /* len(encoderLayerSizes) (Learner_approximator::createEncoder, Learner_approximator.cpp:149-166):  */
  int32_t encoder[HL_MAX_HIDDEN];    /*   dense layers of the preprocessing network, in front of nnLayerSizes in the same network
                                          (Approximator::buildPreprocessing, Approximator.cpp:231-271); n_encoder + n_hidden <= HL_MAX_HIDDEN */
  int32_t encoder_rnn;               /* 1: the encoder layers are plain recurrent layers ("RNN", Builder.cpp:76-81) whatever nn_type says -- what a
                                          partially observable MDP gets for them when nnType is left non-recurrent (Approximator.cpp:264-270),
                                          under the MGU layers of :221-223.  Only with nn_type == HL_NN_MGU. */
} hl_config;

typedef struct hl_learner hl_learner;  /* opaque */

/* scalars of the learner state (device-resident in the HIP library) */
typedef struct hl_scalars {
  double beta, alpha, CmaxRet, CinvRet;            /* MemoryBuffer.h:41-44                 */
  int64_t nGradSteps, nStoredSteps, nStoredEps;
  int64_t nFarPolicySteps;                         /* ReplayStats::nFarPolicySteps          */
  int64_t nSeenSteps, nSeenEps;                    /* ReplayCounters::nSeenTransitions / nSeenEpisodes: summed over the replicas, as of the
                                                      last updateCounters (MemoryProcessing.cpp:60-61) -- what MemoryBuffer::getMetrics
                                                      prints; this replica's live counters: hl_get_counts (the reference's *_loc) */
  double adam_beta_t_1, adam_beta_t_2;             /* Optimizer.h:96                        */
  int64_t adam_nStep;
} hl_scalars;

/* replay statistics as reduced by MemoryProcessing::updateTrainingStatistics (:236-258) */
typedef struct hl_stats {
  double avgKLdivergence, avgSquaredErr, maxAbsError, avgReturn, avgQ, stdevQ, minQ, maxQ;
  int64_t nFarPolicySteps;
  /* ReplayStats::countReturnsEstimateUpdates / sumReturnsEstimateErrors (:253-258): estimates rewritten by the 1000-step
   * sweeps since the last statistics line and the sum of their squared changes -- the "dRet" column; -1 / 0 once printed */
  int64_t countReturnsEstimateUpdates;
  double sumReturnsEstimateErrors;
} hl_stats;

/* per-sample / per-step taps of the LAST executed step (hl_readback) */
enum {
  HL_TAP_FLAT = 0,      /* int64[B]   sampled flat transition indices (sorted)            */
  HL_TAP_EPISODE = 1,   /* int64[B]   episode position in the current episode order       */
  HL_TAP_TSTEP = 2,     /* int64[B]   step within episode                                  */
  HL_TAP_TAG = 3,       /* int64[B]   caller-supplied episode tag                          */
  HL_TAP_STATE = 4,     /* f32[B*dS]  standardized states (MiniBatch::S)                   */
  HL_TAP_OUTPUT = 5,    /* f64[B*nOut] network outputs O                                   */
  HL_TAP_OUTGRAD = 6,   /* f64[B*nOut] output gradient placed by RACER::Train              */
  HL_TAP_RHO = 7,       /* f64[B]                                                          */
  HL_TAP_DKL = 8,       /* f64[B]                                                          */
  HL_TAP_DELTAQ = 9,    /* f64[B]                                                          */
  HL_TAP_FAR = 10,      /* u8[B]     ReF-ER far-policy mask                                */
  HL_TAP_GRADSUM = 11   /* f32[nParams] summed weight gradient before Adam (local rank)   */
};

/* per-step episode fields (hl_get_episode_field) */
enum { HL_EP_RETURN = 0, HL_EP_VALUE = 1, HL_EP_ADVANTAGE = 2, HL_EP_IMPW = 3, HL_EP_DKL = 4,
       HL_EP_DELTAQ = 5 };

/* Threading: every entry point takes the learner's own lock, so calls on one handle may come from several threads -- the
 * training thread stepping while env-service threads hand over finished episodes (hl_append_episode, which never waits
 * for the device) or ask for actions (hl_forward); the reference's dataset_mutex (ReplayMemory/MemoryBuffer.h:55,
 * callers Core/Master.cpp:66-86).  Appended episodes are visible to every later call.  hl_destroy must not race with
 * other calls on the same handle. */

/* ---- lifetime ------------------------------------------------------------ */
HL_API int hl_create(const hl_config* cfg, hl_learner** out);
HL_API int hl_destroy(hl_learner* h);
HL_API const char* hl_last_error(const hl_learner* h);   /* never NULL */
HL_API const char* hl_status_string(int status);
HL_API int hl_version(void);

/* ---- network parameters ---------------------------------------------------- */
HL_API int64_t hl_num_params(const hl_learner* h);        /* padded blob length (72 976 @ cfg-NS) */
HL_API int32_t hl_num_outputs(const hl_learner* h);       /* 1 + 2*dA for VRACER */
HL_API int32_t hl_num_layers(const hl_learner* h);
/* per layer l: offset/length of W and b inside the padded blob (Parameters::indWeights etc.) */
HL_API int hl_param_layout(const hl_learner* h, int64_t* indW, int64_t* nW, int64_t* indB, int64_t* nB);
HL_API int hl_init_weights(hl_learner* h);                /* draws from the learner's mt19937 */
HL_API int hl_set_params(hl_learner* h, const float* w, const float* m1, const float* m2); /* NULL = keep */
HL_API int hl_get_params(hl_learner* h, float* w, float* m1, float* m2);                   /* NULL = skip */

/* ---- sampler RNG: std::mt19937 generators[0] (ExecutionInfo.cpp:391) ------- */
HL_API int hl_set_rng_state(hl_learner* h, const uint32_t state[625]); /* 624 words + position */
HL_API int hl_get_rng_state(hl_learner* h, uint32_t state[625]);

/* ---- replay ---------------------------------------------------------------- */
/* One finished episode of nsteps states.  actions/mu/rewards are f64 as in
 * Episode.h:73-74 (last action/mu row = zeros, rewards[0] = 0), values/advantages
 * are the behaviour-time V(s_t) and A(s_t,a_t) (advantages may be NULL = 0).
 * The library computes the Retrace estimate on insert (MemoryProcessing.cpp:452-458). */
HL_API int hl_append_episode(hl_learner* h, int32_t nsteps, const float* states,
                             const double* actions, const double* mu, const double* rewards,
                             const float* values, const float* advantages,
                             int32_t terminated, int64_t tag);
HL_API int hl_get_scaling(hl_learner* h, float* stateMean, float* stateScale, float* rew3 /*mean,scale,std*/);
HL_API int hl_set_scaling(hl_learner* h, const float* stateMean, const float* stateScale, const float* rew3);
HL_API int hl_get_episode_field(hl_learner* h, int64_t episode_pos, int32_t field, float* dst, int32_t cap);
HL_API int hl_get_episode_info(hl_learner* h, int64_t episode_pos, int64_t* tag, int32_t* nsteps, int32_t* terminated);
/* the episode's running aggregates (Episode.h:82-85, the members ERoldSeqFilter and the statistics pass read), in this order:
 * totR, avgKLDivergence, fracFarPolSteps, avgSquaredErr, maxAbsError, sumSquaredQ, sumQ, maxQ, minQ */
HL_API int hl_get_episode_stats(hl_learner* h, int64_t episode_pos, float* dst9);

/* ---- training ---------------------------------------------------------------- */
HL_API int hl_initialize(hl_learner* h);                  /* Learner::initializeLearner */
/* ... in two halves for host-exchange mode (n_ranks > 1, the caller owns the communicator): the reference's start-up reductions are
 * accurate ones (updateCounters(true), updateRewardsStats(true): DelayedReductor::get(true) waits, Learner.cpp:58-59), every learner
 * starts from the GLOBAL counters and reward / state moments.  Between the halves the caller sums hl_counters_exchange and
 * hl_moments_exchange over the replicas (smarties_amd/dist_host.py: initialize_host_exchange).  hl_initialize = both halves with
 * the exchange over the library's own communicator (hl_xchg_connect / hl_comm_init) in between. */
HL_API int hl_initialize_begin(hl_learner* h);
HL_API int hl_initialize_end(hl_learner* h);
/* n_steps full gradient steps.  flat_indices == NULL: device-side sampler; else
 * n_steps * batch_local sorted unique indices to use instead (the RNG is then
 * advanced as if it had drawn them only by the per-step Adam draw). */
HL_API int hl_step(hl_learner* h, int32_t n_steps, const int64_t* flat_indices);
/* split form: begin = sample + train + local gradient sum; end = bookkeeping + Adam.
 * Between the two the caller may all-reduce hl_grad_host() itself (host MPI/gloo path). */
HL_API int hl_step_begin(hl_learner* h, const int64_t* flat_indices);
HL_API int hl_grad_exchange(hl_learner* h, float* grad_io /*nParams, NULL = fetch only into internal host buf*/, int32_t write_back);
HL_API int hl_counters_exchange(hl_learner* h, int64_t counters_io[4], int32_t write_back);
/* every 1000th step: the 2*dS+3 reward/state moments of MemoryProcessing.cpp:139-150 (C3) */
HL_API int hl_moments_exchange(hl_learner* h, double* io, int32_t write_back);
HL_API int hl_step_end(hl_learner* h);
HL_API int hl_sync(hl_learner* h);                         /* wait for all queued device work */
/* A caller that steps n gradient steps per hl_step call -- the reference's training task does one per turn
 * (Learners/RACER.cpp:81-108), a throughput loop many -- may announce n once after hl_initialize: the replayed graph of exactly
 * n steps is built here instead of being assembled from the stock sizes (20 = 16 + 4), and its last node stamps a pinned host
 * word that hl_sync polls.  Purely an optimisation: results are bit-identical with or without it; call sizes seen three
 * times in a row are prepared automatically. */
HL_API int hl_prepare_steps(hl_learner* h, int32_t n_steps);

/* ---- rollout inference (first row) --------------------------------------
 * Network outputs for n raw (un-standardised) states with the CURRENT weights and state scaling:
 * what Approximator::forward(agent) returns to RACER::selectAction / processTerminal
 * (Learners/RACER.cpp:30-59; Network/Approximator.h:300-330): outputs[i] = [V_net, mean[dA],
 * sigma_param[dA]] as doubles, nOut per state.  The caller builds the policy, draws the action
 * with the agent's generator and applies scaleNet2V exactly as the reference does.  Call between
 * steps, from the thread that owns the learner. */
HL_API int hl_forward(hl_learner* h, int32_t n, const float* states /*[n][dimS]*/, double* outputs /*[n][nOut]*/);

/* ---- episodes in the reference's wire format (second row) --------------------
 * Episode::packEpisode / unpackEpisode (ReplayMemory/Episode.cpp:24-130): the fp32 record a worker
 * sends to the learner and MemoryBuffer::save writes per episode.  nsteps * (dimS + 1 + dimA +
 * 2 dimA + 6) floats -- per step [state | reward | action | policy], then returnEstimator,
 * actionAdvantage, stateValue, deltaValue, offPolicImpW, KullbLeibDiv -- plus 10 floats holding
 * {bool bReachedTermState, Sint ID, Sint just_sampled, Sint agentID} byte-packed (Episode.h:211-219).
 * hl_append_packed_episode = unpackEpisode + the hl_append_episode path (actions, policies and
 * rewards come back from fp32 exactly as the reference reads them; ID becomes the episode tag).
 * hl_pack_episode packs a STORED episode with its current derived fields. */
HL_API int64_t hl_packed_episode_size(const hl_learner* h, int32_t nsteps);
HL_API int hl_append_packed_episode(hl_learner* h, const float* data, int64_t n_floats);
HL_API int hl_pack_episode(hl_learner* h, int64_t episode_pos, float* dst, int64_t cap_floats);

/* ---- checkpoint in the reference's file format (second row) ----------------
 * Approximator::save -> AdamOptimizer::save -> Network::save (Network/Approximator.cpp:282-297,
 * Network/Optimizer.cpp:180-214, Network/Network.cpp:22-68): three raw fp32 files
 * <base>_weights.raw, <base>_1stMom.raw, <base>_2ndMom.raw, layer by layer WITHOUT the SIMD padding
 * of the in-memory blob (dense: W[in][out] then bias; parametric residual: w then b; ParamLayer:
 * bias).  The binding passes base = "<agent>_net" ("agent_00_net").  hl_restart needs the weights
 * file (HL_ERR_IO if it is missing or has the wrong size); missing moment files are ignored, as in
 * AdamOptimizer::restart. */
/* Acting with recurrent layers (MemoryBuffer::agentToMinibatch, MemoryBuffer.cpp:440-467 + Approximator::forward(agent)):
 * `states` = the agent's last n_steps raw observed states, oldest first, n_steps = min(nnBPTTseq, t) + 1; every one is
 * forwarded from a zero recurrent state, `outputs` (nOutputs doubles) are those of the last.  Dense nets: only the last
 * state matters (same result as hl_forward on it).  Recurrent layers behind appended observations: up to nAppendedObs further
 * states may stand in front of the window (n_steps <= nnBPTTseq + 1 + nAppendedObs); they only fill the appended slots of the
 * window's first steps, and steps before the first given state repeat it (Episode::standardizedState, Episode.h:172-183). */
HL_API int hl_forward_sequence(hl_learner* h, int32_t n_steps, const float* states, double* outputs);
HL_API int hl_save(hl_learner* h, const char* base);
HL_API int hl_restart(hl_learner* h, const char* base);

/* Replay memory + ReF-ER state in the reference's files (MemoryBuffer::save / restart,
 * ReplayMemory/MemoryBuffer.cpp:172-324): <base>_scaling.raw (doubles: state mean | scale | stdev,
 * reward stdev, scale, mean), <base>_rank_RRR_learner_status.raw (text: nStoredEps, nStoredObs,
 * nLocalSeenEps, nLocalSeenObs, nInitialData, nGradSteps, CmaxReFER, beta) and
 * <base>_rank_RRR_learner_data.raw (per episode: Uint length + Episode::packEpisode record, with every
 * derived per-step field).  hl_restart_memory restores the per-step fields as stored, recomputes the
 * per-episode aggregates (Episode::updateCumulative) and -- like the reference, which skips
 * Learner::initializeLearner for a restarted learner -- leaves the learner ready to step.
 * Missing files: HL_ERR_IO (the reference prints a notice and continues with an empty memory). */
HL_API int hl_save_memory(hl_learner* h, const char* base, int32_t rank);
HL_API int hl_restart_memory(hl_learner* h, const char* base, int32_t rank);

/* ---- inspection ---------------------------------------------------------------- */
HL_API int hl_set_tap(hl_learner* h, int32_t enable);
HL_API int hl_readback(hl_learner* h, int32_t what, void* dst, int64_t dst_bytes);
HL_API int hl_get_scalars(hl_learner* h, hl_scalars* out);
/* the host-side counters only -- no device wait: what Learner::locDataSetSize / nGradSteps / nLocTimeSteps read between
 * steps (Learners/Learner.h:84-101).  Any pointer may be NULL. */
HL_API int hl_get_counts(hl_learner* h, int64_t* nStoredSteps, int64_t* nStoredEps, int64_t* nGradSteps, int64_t* nSeenSteps, int64_t* nSeenEps);
HL_API int hl_get_stats(hl_learner* h, hl_stats* out);
/* ReplayCounters::nGatheredB4Startup ("nInitialData" of the status file, MemoryBuffer.cpp:249-250, 302-311): the locally seen
 * observations from which Learner::nLocTimeStepsTrain counts -- set by hl_initialize, restored by hl_restart_memory; INT64_MAX
 * before either.  A binding that keeps the reference's gating (blockGradientUpdates, Learner.cpp:116-123) copies it after a restart. */
HL_API int hl_get_initial_data(hl_learner* h, int64_t* nInitialData);

/* ---- statistics surface (third row) ---------------------------------------
 * The column header and the line Learner::logStats appends to <learner>_stats.txt
 * (Learners/Learner.cpp:155-195): MemoryBuffer::getHeaders / getMetrics (MemoryBuffer.cpp:522-575),
 * then the network's AdamOptimizer::getHeaders / getMetrics (Optimizer.cpp:216-226), formatted with
 * Utilities::real2SS (Utils/SstreamUtilities.h:51-63).  Either buffer may be NULL. */
HL_API int hl_metrics(hl_learner* h, char* header, int32_t header_cap, char* line, int32_t line_cap);

/* The histogram of the off-policy importance weights Learner::logStats prints with the profiler every freqPrint x
 * PRFL_DMPFRQ steps (MemoryProcessing::histogramImportanceWeights, ReplayMemory/MemoryProcessing.cpp:353-389; caller
 * Learners/Learner.cpp:139-144): 81 bins over the stored transitions' pi/mu -- [0, 1e-3), 79 log-spaced bins up to 50,
 * [50, max) --, counted on the device.  `text` (may be NULL) receives the block exactly as the reference prints it,
 * `counts` (may be NULL) the 81 bin counts. */
#define HL_IMPW_BINS 81
HL_API int hl_impweight_histogram(hl_learner* h, char* text, int32_t text_cap, int64_t counts[HL_IMPW_BINS]);

/* <runDir>/agent_XX_rank_RRR_cumulative_rewards.dat (MemoryBuffer::pushBackEpisode, ReplayMemory/MemoryBuffer.cpp:481-507,
 * with --logAllSamples): one line "nGradSteps timeStamp agentID nSteps cumulativeReward" per episode that enters the
 * training set, appended to `path` (NULL / "" switches it off).  The companion _obs.raw holds environment-scaled actions
 * and latent state variables (Episode::logToFile): that one is written where those are known -- the reference's own
 * MemoryBuffer in the compiled binding (bindings/smarties/RACER_HIP.h keeps it as the episode collector). */
HL_API int hl_set_episode_log(hl_learner* h, const char* path);

/* Output-gradient statistics (Utils/StatsTracker.cpp:28-107, fed by Approximator::setGradient,
 * Network/Approximator.h:197): mean and root-mean-square over the last minibatch of each network
 * output's gradient (nOutputs values each).  hl_set_log_base(h, "<learner_name>") makes hl_step /
 * hl_step_end append them to <learner_name>_net_outGrad_stats.raw for the steps with
 * nGradSteps % 1000 == 0, in the reference's format (first a float nOutputs + 0.1, then 2*nOutputs floats
 * per record; rank 0 only), as Learner_approximator.cpp:89 does.  NULL / "" switches the file off. */
HL_API int hl_grad_stats(hl_learner* h, double* mean, double* rms);
HL_API int hl_set_log_base(hl_learner* h, const char* base);

/* ---- multi-GPU (RCCL over xGMI) -------------------------------------------------- */
HL_API int hl_comm_unique_id(uint8_t id[128]);             /* rank 0 creates, caller broadcasts */
HL_API int hl_comm_init(hl_learner* h, const uint8_t id[128]);
/* The same sums without RCCL (2..16 replicas of one node): every replica owns a window in its HBM into which each peer writes
 * its message directly over xGMI; ONE kernel per collective stores the local message into all peers' windows, waits for theirs
 * and sums in rank order (bit-identical replicas) -- replacing the reference's MPI_Iallreduce of the gradient
 * (Network/Optimizer.cpp:110-132) and of the counters / moments (Utils/DelayedReductor.cpp:53-83).  hl_xchg_export creates the
 * window and returns its handle; the caller gathers all replicas' handles IN RANK ORDER with its own communicator (the
 * reference's learners have MPI: one MPI_Allgather of HL_XCHG_HANDLE_BYTES bytes) and hands them to hl_xchg_connect, which maps
 * the peers' windows (hipIpc between processes, plain pointers between learners of one process), copies rank 0's weights to
 * every replica like hl_comm_init and makes hl_initialize / hl_step use this exchange (it takes precedence over a
 * communicator of hl_comm_init).  A peer that does not answer within SMARTIES_HIP_XCHG_TIMEOUT_MS (default 5000) raises the
 * learner's device error instead of hanging the GPU.  Replicas of ONE process wait for each other inside kernels, so each needs a
 * hardware queue of its own: HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues -- raise it for more replicas per
 * process; one process per GPU, the usual layout, is not affected. */
#define HL_XCHG_HANDLE_BYTES 96
HL_API int hl_xchg_export(hl_learner* h, uint8_t handle[HL_XCHG_HANDLE_BYTES]);
HL_API int hl_xchg_connect(hl_learner* h, const uint8_t* handles /* n_ranks x HL_XCHG_HANDLE_BYTES */);
/* n_ranks > 1 WITHOUT hl_xchg_connect / hl_comm_init = host-exchange mode: the caller owns the communicator and
 * drives hl_step_begin / hl_grad_exchange / hl_counters_exchange / hl_moments_exchange /
 * hl_step_end itself, and hl_initialize_begin / hl_counters_exchange / hl_moments_exchange / hl_initialize_end at start-up
 * (smarties_amd/dist_host.py); a plain hl_initialize takes the start-up reward / state statistics from the local shard only, and
 * hl_step returns HL_ERR_COMM. */

/* ---- timing taps for bench.py ------------------------------------------------------ */
/* average device time (ms) per launch of the named kernel over the launches since the
 * last hl_timing_reset, measured with HIP events on the library's own stream */
HL_API int hl_timing_enable(hl_learner* h, int32_t enable);
HL_API int hl_timing_get(hl_learner* h, const char* kernel, double* avg_ms, int64_t* launches);

/* Isolated kernel profile: captures `reps` back-to-back launches of ONE kernel of the step into a
 * graph on the library's stream, replays it 20 times between two HIP events and returns the
 * average microseconds per launch (this includes the ~1.6 us graph-node dispatch gap; profile
 * HL_PROF_EMPTY the same way to calibrate it).  The 2x launches are issued exactly as inside the
 * replayed step, i.e. with the horizontally fused sampler / bookkeeping workgroup riding along.
 * The call advances the sampler state and re-applies updates: profile AFTER the timed run. */
enum {
  HL_PROF_SAMPLE = 0,      /* stand-alone sampler (first minibatch of a replayed graph) */
  HL_PROF_EMPTY = 12,      /* empty kernel node: dispatch-gap calibration */
  HL_PROF_STEP_GRAPH = 7,  /* `reps` whole steps as one replayed graph */
  HL_PROF_FWD0 = 21,       /* first forward GEMM   + sampler phase A of the next step */
  HL_PROF_FWD_LAST = 22,   /* last forward GEMM    + sampler phase B */
  HL_PROF_HEAD = 23,       /* V-RACER head         + sampler phase C (search + gather) */
  HL_PROF_DX = 24,         /* first backward dX GEMM + ReF-ER bookkeeping of this step */
  HL_PROF_DW = 25,         /* all dW GEMMs + bias reductions + fused Adam */
  /* networks served by the fused kernel (two equal hidden blocks): the two launches of a replayed step */
  HL_PROF_FUSED = 26,          /* forward + head + dX, sampler of the next step riding along */
  HL_PROF_FUSED_DW = 27,       /* all weight gradients + Adam, bookkeeping of the step riding along */
  HL_PROF_FUSED_BARE = 28,     /* 26 without its rider */
  HL_PROF_FUSED_DW_BARE = 29   /* 27 without its rider */
};
HL_API int hl_kernel_profile(hl_learner* h, int32_t which, int32_t reps, double* us_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SMARTIES_HIP_H */
