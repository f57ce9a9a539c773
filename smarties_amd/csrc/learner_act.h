// smarties_amd/csrc/learner_act.h -- part of learner.cpp's ONE translation unit (included there, like step_exec.h): rollout inference (Learner::select's network evaluation): hl_forward, hl_forward_sequence
#pragma once

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
