// smarties_amd/csrc/learner.cpp -- host side of libsmarties_hip.so: the hl_* C-ABI of
// include/smarties_hip.h.  Owns the device-resident replay buffer, the parameter blob, the
// per-step launch sequence (eager or as a replayed hipGraph) and the RCCL communicator.
// There is no CPU compute path in this library: every entry point that needs the GPU fails
// with HL_ERR_NO_DEVICE / HL_ERR_HIP if HIP is not usable.
//
// ONE translation unit in seven files (round 6; the launch lists were step_exec.h already): learner_state.h (struct hl_learner, helpers,
// network description), step_exec.h (launch lists, graphs, exchanges), this file (create / destroy, parameters, ingestion, initialisation,
// the stepping entry points, read-backs), learner_io.h (wire format, metrics, checkpoints), learner_act.h (rollout inference),
// learner_xchg.h (communicator and exchange windows), learner_debug.h (timing and development entry points).
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

#include "learner_state.h"
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
#include "learner_io.h"
#include "learner_act.h"

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

#include "learner_xchg.h"
}  // extern "C"
#include "learner_debug.h"
