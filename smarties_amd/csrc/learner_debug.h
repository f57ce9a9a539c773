// smarties_amd/csrc/learner_debug.h -- part of learner.cpp's ONE translation unit (included there, like step_exec.h): timing taps and development entry points (hl_timing_*, hl_kernel_profile, hl_debug_*)
#pragma once

extern "C" {
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
