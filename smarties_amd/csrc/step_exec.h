// smarties_amd/csrc/step_exec.h -- launch sequences of the gradient step (included by learner.cpp).
//
// One step = sample -> forward -> head -> dX -> dW (+Adam) -> bookkeeping.  The sampler and the
// bookkeeping pass are single-workgroup dependency chains (~10 us and ~3.5 us) that do not touch
// what the MLP kernels touch; in the replayed graph they ride along the MLP kernels as extra
// workgroups (tail_dev.h).
//
// Networks with two equal hidden blocks (fusedOk) take two launches per step:
//   K1 fused_fwd_head_dx(k)  + block 0: sampler of step k+1 (draw, sort/unique, search),
//                              blocks 1..7: its gather                           (fused.hip)
//   K2 dw_table(k) + Adam    + block 0: bookkeeping of step k (aggregates, beta/C, Adam scalars)
// every other layout the generic five:
//   fwd0(k) + sampler phase A | fwd_last(k) + phase B | head(k) + phase C | dX(k) + bookkeeping | dW(k)
// With a communicator attached (replicas) K2 leaves Adam out and is followed by
//   allreduce(G), allreduce(counters), adam, bookkeeping(beta)   -- RCCL calls captured like kernels.
//
// Step k reads minibatch buffer k&1 while step k+1's is being written, so the minibatch
// workspace (indices, standardized states, row counts, Adam step size) is double buffered;
// everything else is single buffered.  (A multi-stream graph with the same overlap was measured
// 1.5x SLOWER than the serial graph on this runtime: each cross-queue edge costs several us.)
// Eager launches (parity tests, 1000-step sweeps, eviction) run the same kernels back to back on
// the main stream with stand-alone tail launches.
#pragma once

namespace {

// steps per replayed graph, tried in this order (a call of n steps is served greedily: 20 = 16 + 4);
// 999 = everything between two 1000th-step sweeps
constexpr int GRAPH_SIZES[] = {999, 256, 128, 64, 32, 16, 8, 4, 2, 1};
constexpr int CNT_MSG_FLOATS = 16;     // four counters x four 16-bit chunks (tail_dev.h: postPart)
constexpr int CNT_MSG_OFFSET = 128;    // counters message inside the PARAM_TAIL floats behind the gradient (learner.cpp)

void setTiles(GemmProblem& p, int& cursor, bool maySplit = false) {
  p.tilesM = (p.M + 15) / 16; p.tilesN = (p.N + 15) / 16;
  if (p.flavor == RED_COL) { p.tilesM = 1; p.tilesN = (p.N + 15) / 16; }
  // weight gradients over >= 1024 rows (recurrent nets: batch x BPTT steps): one workgroup per (tile, 256-row chunk)
  p.nSplit = (maySplit && (p.flavor == GEMM_W || p.flavor == RED_COL) && p.K >= 1024) ? (p.K + 255) / 256 : 1;
  p.tileStart = cursor; cursor += p.tilesM * p.tilesN * p.nSplit;
}

// GEMM problem tables, one set per minibatch buffer (they differ in X0 / gParam only)
int buildProblems(hl_learner* h) {
  std::vector<GemmProblem> P;
  const int nH = h->nHidden, B = h->B;
  for (int pb = 0; pb < 2; ++pb) {
    StepBuf& sb = h->buf[pb];
    sb.fwdIdx.clear(); sb.fwdBlocks.clear(); sb.dxIdx.clear(); sb.dxBlocks.clear(); sb.bigDw.clear();
    // forward: one launch per hidden block (dense layers; LSTM layers have their own kernels, rec.hip; with convolutional
    // preprocessing hid[0] is the last convolution, computed by conv.hip)
    const int j0 = h->nConv > 0 ? 1 : 0;
    if (j0) { sb.fwdIdx.push_back(-1); sb.fwdBlocks.push_back(0); }
    for (int j = j0; j < nH && !h->recurrent; ++j) {
      const DevHidden& d = h->hid[j];
      GemmProblem p{}; p.flavor = GEMM_F; p.epi = EPI_FWD; p.M = h->Mmax; p.N = d.size; p.K = d.nIn; p.dynRows = 1;
      if (j == 0) { p.A = sb.X0; p.lda = h->ldX0; }
      else { const DevHidden& q = h->hid[j - 1]; p.A = q.hasRes ? q.Rr : q.Y; p.lda = q.ldA; }
      p.B = h->W + d.indW; p.ldb = d.ldW; p.bias = h->W + d.indB;
      p.C = d.X; p.C2 = d.Y; p.C3 = d.hasRes ? d.Rr : nullptr; p.ldc = d.ldA; p.func = d.func;
      if (d.hasRes) { p.resW = h->W + d.indWr; p.resB = h->W + d.indBr; p.resIn = p.A; p.ldRes = p.lda; p.resN = d.resW; }
      int cur = 0; setTiles(p, cur);
      sb.fwdIdx.push_back((int)P.size()); sb.fwdBlocks.push_back(cur); P.push_back(p);
    }
    // dX: block j = nH-1 .. 1: Dres_{j-1} = D_j W_j^T + Dres_j[:, :res] * w_j ; D_{j-1} = Dres_{j-1} * act'
    for (int j = nH - 1; j >= 1 && !h->recurrent; --j) {
      const DevHidden& d = h->hid[j]; const DevHidden& q = h->hid[j - 1];
      GemmProblem p{}; p.flavor = GEMM_X; p.epi = EPI_DX; p.M = B; p.N = d.nIn; p.K = d.size;
      p.A = d.D; p.lda = d.ldA; p.B = h->W + d.indW; p.ldb = d.ldW;
      p.C = q.Dres; p.C2 = q.D; p.ldc = q.ldA;
      if (d.hasRes) { p.resIn = d.Dres; p.ldRes = d.ldA; p.resW = h->W + d.indWr; p.resN = d.resW; }
      p.actX = q.X; p.actY = q.Y; p.ldAct = q.ldA; p.func = q.func;
      int cur = 0; setTiles(p, cur);
      sb.dxIdx.push_back((int)P.size()); sb.dxBlocks.push_back(cur); P.push_back(p);
    }
    // dW: every weight / bias / residual-parameter gradient in one multi-problem launch
    // two layer types (hl_config::encoder_rnn): the error of the lower segment's outputs, per window row = (gate deltas of the upper
    // segment's first layer) W_in^T + its residual path (C2 of the dX epilogue, the product with an activation's derivative, is not
    // used here: a linear one into scratch)
    if (h->recurrent && h->recSplit) {
      const int ju = j0 + h->recSplit; const DevHidden& d = h->hid[ju]; const RecLayer& L = h->rec[ju];
      const int NO = std::max(d.lstm, 1) * d.size;
      GemmProblem p{}; p.flavor = GEMM_X; p.epi = EPI_DX; p.M = B * h->recK; p.N = d.nIn; p.K = NO;
      p.A = L.D; p.lda = NO; p.B = h->W + d.indW; p.ldb = d.ldW;
      p.C = h->segDres; p.C2 = h->segScratch; p.ldc = h->ldSeg;
      if (d.hasRes) { p.resIn = L.Rd; p.ldRes = L.ldR; p.resW = h->W + d.indWr; p.resN = d.resW; }
      p.actX = h->segY; p.actY = h->segY; p.ldAct = h->ldSeg; p.func = HL_FUNC_LINEAR;
      int cur = 0; setTiles(p, cur);
      sb.segDxIdx = (int)P.size(); sb.segDxBlocks = cur; P.push_back(p);
    }
    // convolutions in front of recurrent layers: Dres of the last convolution's rows = (gate deltas of the first recurrent layer)
    // W_in^T + the residual path, over the B K window rows (the deltas of the gates are what rec_backward left for the dW launch)
    if (h->recurrent && j0) {
      const DevHidden& d = h->hid[1]; const DevHidden& q = h->hid[0]; const RecLayer& L = h->rec[1];
      const int NO = std::max(d.lstm, 1) * d.size;
      GemmProblem p{}; p.flavor = GEMM_X; p.epi = EPI_DX; p.M = h->convB; p.N = d.nIn; p.K = NO;
      p.A = L.D; p.lda = NO; p.B = h->W + d.indW; p.ldb = d.ldW;
      p.C = q.Dres; p.C2 = q.D; p.ldc = q.ldA;
      if (d.hasRes) { p.resIn = L.Rd; p.ldRes = L.ldR; p.resW = h->W + d.indWr; p.resN = d.resW; }
      p.actX = q.X; p.actY = q.Y; p.ldAct = q.ldA; p.func = q.func;
      int cur = 0; setTiles(p, cur);
      sb.dxIdx.push_back((int)P.size()); sb.dxBlocks.push_back(cur); P.push_back(p);
    }
    sb.dwIdx = (int)P.size(); int cur = 0;
    // large batches: every weight-gradient problem -- 64 x 64 tiles of the products, 64-column blocks of the column sums -- over row
    // chunks in ONE launch (bigmm.hip: big_dw_kernel), joined by splitk_reduce; none of their tiles in the common launch
    auto placeDw = [&](GemmProblem& p, bool maySplit) {
      const int chunk = big_dw_chunk_rows(p);
      if ((h->bigMm & 2) && (int)sb.bigDw.size() < BIG_DW_MAX && big_dw_ok(p) && p.K > chunk) {      // (two chunks at least: the join owns the gradient and Adam)
        p.bigChunk = chunk; p.nSplit = (p.K + chunk - 1) / chunk;
        p.tilesM = 0; p.tilesN = 0; p.tileStart = cur; sb.bigDw.push_back((int)P.size());
      } else setTiles(p, cur, maySplit);
    };
    for (int j = j0; j < nH && h->recurrent; ++j) {
      // LSTM layer: gradient of [W_in; W_rec] and of the bias as X^T delta over all (sample, step) rows; rows of steps a
      // sample does not have carry zero deltas (rec_backward_kernel)
      const RecLayer& L = h->rec[j]; const int R = B * h->recK;
      if (h->hid[j].lstm == 1) {   // dense layer with a recurrent term: rows [input | previous output | 1], one delta column block
        GemmProblem p{}; p.flavor = GEMM_W; p.epi = EPI_DW; p.M = L.nIn + L.nC + 1; p.N = L.nC; p.K = R;
        p.A = L.A; p.lda = L.ldA; p.B = L.D; p.ldb = L.nC; p.C = h->G + L.indW; p.ldc = h->hid[j].ldW; p.biasOut = h->G + L.indB;
        placeDw(p, true); P.push_back(p);
      } else if (h->hid[j].lstm == 4) {
        GemmProblem p{}; p.flavor = GEMM_W; p.epi = EPI_DW; p.M = L.nIn + L.nC + 1; p.N = 4 * L.nC; p.K = R;
        p.A = L.A; p.lda = L.ldA; p.B = L.D; p.ldb = 4 * L.nC; p.C = h->G + L.indW; p.ldc = 4 * L.nC; p.biasOut = h->G + L.indB;
        placeDw(p, true); P.push_back(p);
      } else {
        // MGU (Layer_GRU.h:196-229): [Wff Wsf] and the biases from the inputs; Wfr from the previous output and dLdF; Wsr from
        // (previous output x forget) and dLdS.  The two recurrent blocks have no bias: their bias row goes to the unused tail
        // of the parameter arrays.
        const int nC = L.nC;
        GemmProblem p{}; p.flavor = GEMM_W; p.epi = EPI_DW; p.M = L.nIn + 1; p.N = 2 * nC; p.K = R;
        p.A = L.A; p.lda = L.ldA; p.B = L.D; p.ldb = 2 * nC; p.C = h->G + L.indW; p.ldc = 2 * nC; p.biasOut = h->G + L.indB;
        placeDw(p, true); P.push_back(p);
        GemmProblem f{}; f.flavor = GEMM_W; f.epi = EPI_DW; f.M = nC + 1; f.N = nC; f.K = R;
        f.A = L.A + L.nIn; f.lda = L.ldA; f.B = L.D; f.ldb = 2 * nC; f.C = h->G + L.indW + (long long)2 * nC * L.nIn; f.ldc = 2 * nC;
        f.biasOut = h->G + h->nParams;
        placeDw(f, true); P.push_back(f);
        GemmProblem q{}; q.flavor = GEMM_W; q.epi = EPI_DW; q.M = nC + 1; q.N = nC; q.K = R;
        q.A = L.A2; q.lda = L.ldA2; q.B = L.D + nC; q.ldb = 2 * nC; q.C = h->G + L.indW + (long long)2 * nC * L.nIn + nC; q.ldc = 2 * nC;
        q.biasOut = h->G + h->nParams + 64;
        placeDw(q, true); P.push_back(q);
      }
      if (L.hasRes) {
        GemmProblem r{}; r.flavor = RED_COL; r.epi = EPI_NONE; r.N = L.resW; r.K = R;
        r.A = L.Rd; r.lda = L.ldR; r.B = L.A; r.ldb = L.ldA; r.C = h->G + L.indWr;
        placeDw(r, true); P.push_back(r);
        GemmProblem s2{}; s2.flavor = RED_COL; s2.epi = EPI_NONE; s2.N = L.resW; s2.K = R;
        s2.A = L.Rd; s2.lda = L.ldR; s2.B = nullptr; s2.C = h->G + L.indBr;
        placeDw(s2, true); P.push_back(s2);
      }
    }
    for (int l = 0; l < h->nConv; ++l) {   // convolution biases (one per output element): column sums of the layer's deltas
      const ConvGeo& g = h->cg[l];
      GemmProblem s2{}; s2.flavor = RED_COL; s2.epi = EPI_NONE; s2.N = g.KnC * g.P; s2.K = h->convB;
      s2.A = g.D; s2.lda = g.ldOut; s2.B = nullptr; s2.C = h->G + g.indB;
      setTiles(s2, cur); P.push_back(s2);
    }
    for (int j = j0; j < nH && !h->recurrent; ++j) {
      const DevHidden& d = h->hid[j];
      GemmProblem p{}; p.flavor = GEMM_W; p.epi = EPI_DW; p.M = d.nIn + 1; p.N = d.size; p.K = B;
      if (j == 0) { p.A = sb.X0; p.lda = h->ldX0; }
      else { const DevHidden& q = h->hid[j - 1]; p.A = q.hasRes ? q.Rr : q.Y; p.lda = q.ldA; }
      p.B = d.D; p.ldb = d.ldA; p.C = h->G + d.indW; p.ldc = d.ldW; p.biasOut = h->G + d.indB;
      placeDw(p, h->bigBatch);      // (large batches off bigmm.hip: one workgroup per (tile, 256-row chunk), as for the recurrent nets' rows)
      P.push_back(p);
      if (d.hasRes) {   // ParametricResidualLayer::backward (Layers.h:363-393)
        GemmProblem r{}; r.flavor = RED_COL; r.epi = EPI_NONE; r.N = d.resW; r.K = B;
        r.A = d.Dres; r.lda = d.ldA; r.B = p.A; r.ldb = p.lda; r.C = h->G + d.indWr;
        placeDw(r, h->bigBatch); P.push_back(r);
        GemmProblem s{}; s.flavor = RED_COL; s.epi = EPI_NONE; s.N = d.resW; s.K = B;
        s.A = d.Dres; s.lda = d.ldA; s.B = nullptr; s.C = h->G + d.indBr;
        placeDw(s, h->bigBatch); P.push_back(s);
      }
    }
    { // output InnerProduct layer
      const DevHidden& q = h->hid[nH - 1];
      GemmProblem p{}; p.flavor = GEMM_W; p.epi = EPI_DW; p.M = q.size + 1; p.N = h->nDense; p.K = B;
      p.A = q.hasRes ? q.Rr : q.Y; p.lda = q.ldA; p.B = h->dOut; p.ldb = h->ldDo;
      p.C = h->G + h->indWo; p.ldc = h->ldWo; p.biasOut = h->G + h->indBo;
      placeDw(p, h->bigBatch); P.push_back(p);
      // ParamLayer::backward (Layers.h:522-546): bias gradient = column sums of the sigma-param deltas
      if (h->nSig) {      // (no sigma layer behind a discrete policy)
        GemmProblem s{}; s.flavor = RED_COL; s.epi = EPI_NONE; s.N = h->dA; s.K = B;
        s.A = sb.bt.gParam; s.lda = h->dA; s.B = nullptr; s.C = h->G + h->indBp;
        placeDw(s, h->bigBatch); P.push_back(s);
      }
    }
    sb.dwCount = (int)P.size() - sb.dwIdx; sb.dwBlocks = cur;
    {   // scratch of the split problems (shared by both minibatch buffers: steps are sequential)
      size_t need = 0; sb.splitMaxMN = 0;
      for (int i = 0; i < sb.dwCount; ++i) { const GemmProblem& q = P[sb.dwIdx + i]; if (q.nSplit > 1) { const int mn = q.flavor == RED_COL ? q.N : q.M * q.N; need += (size_t)q.nSplit * mn; sb.splitMaxMN = std::max(sb.splitMaxMN, mn); } }
      if (need > h->splitPartFloats) { if (h->splitPart) hipFree(h->splitPart); h->splitPart = nullptr; HIPCK(devAlloc(&h->splitPart, need)); h->splitPartFloats = need; }
      size_t off = 0;
      for (int i = 0; i < sb.dwCount; ++i) { GemmProblem& q = P[sb.dwIdx + i]; if (q.nSplit > 1) { q.part = h->splitPart + off; off += (size_t)q.nSplit * (q.flavor == RED_COL ? q.N : q.M * q.N); } }
    }
    // second copy of the dW table with the Adam update fused into the epilogue (single replica:
    // every gradient element is final inside the workgroup that produced it)
    sb.dwAdamIdx = (int)P.size();
    for (int i = 0; i < sb.dwCount; ++i) {
      GemmProblem p = P[sb.dwIdx + i];
      p.adam = p.nSplit > 1 ? 0 : 1; p.adamRed = p.nSplit > 1 ? 1 : 0;
      const long long offC = p.C - h->G;
      p.adW = h->W + offC; p.adM1 = h->M1 + offC; p.adM2 = h->M2 + offC;
      if (p.biasOut) { const long long offB = p.biasOut - h->G; p.adbW = h->W + offB; p.adbM1 = h->M1 + offB; p.adbM2 = h->M2 + offB; }
      P.push_back(p);
    }
    // recurrent nets: third and fourth copy for the one-launch form (gemm16.hip: dw_wide_kernel) -- no row chunks, eight wavefronts
    // per tile over all rows, the Adam update in the tile's epilogue
    sb.dwWideIdx = sb.dwWideAdamIdx = -1; sb.dwWideBlocks = 0;
    if (h->recurrent && sb.splitMaxMN > 0) {
      for (int pass = 0; pass < 2; ++pass) {
        (pass ? sb.dwWideAdamIdx : sb.dwWideIdx) = (int)P.size();
        int curW = 0;
        for (int i = 0; i < sb.dwCount; ++i) {
          GemmProblem q = P[(pass ? sb.dwAdamIdx : sb.dwIdx) + i];
          q.nSplit = 1; q.part = nullptr; q.adamRed = 0; q.adam = pass;
          q.tileStart = curW; curW += q.tilesM * q.tilesN;
          P.push_back(q);
        }
        sb.dwWideBlocks = curW;
      }
      if (sb.dwWideBlocks > h->wideTiles) {      // (shared by both minibatch buffers: steps are sequential)
        if (h->widePart) hipFree(h->widePart); if (h->wideCtr) hipFree(h->wideCtr);
        h->widePart = nullptr; h->wideCtr = nullptr; h->wideTiles = (int)roundUp(sb.dwWideBlocks, 8);
        HIPCK(devAlloc(&h->widePart, (size_t)h->wideTiles * DW_WIDE_Q * 256));
        HIPCK(devAlloc(&h->wideCtr, (size_t)h->wideTiles));
        HIPCK(hipMemset(h->wideCtr, 0, (size_t)h->wideTiles * sizeof(unsigned)));
      }
    }
    sb.dwTable = DwTable{}; sb.dwTableAdam = DwTable{};
    if (sb.dwCount <= DW_TABLE_MAX) {
      sb.dwTable.n = sb.dwTableAdam.n = sb.dwCount;
      for (int i = 0; i < sb.dwCount; ++i) { sb.dwTable.p[i] = P[sb.dwIdx + i]; sb.dwTableAdam.p[i] = P[sb.dwAdamIdx + i]; }
    }
  }
  if (h->dProbs) hipFree(h->dProbs);
  h->dProbs = nullptr;
  HIPCK(devAlloc(&h->dProbs, P.size()));
  HIPCK(hipMemcpy(h->dProbs, P.data(), P.size() * sizeof(GemmProblem), hipMemcpyHostToDevice));
  h->hostProbs = P;
  return HL_OK;
}

AdamHyper adamHyper(const hl_learner* h, int parity) {
  AdamHyper a{}; a.eta0 = (float)h->cfg.learnrate; a.lambda = (float)h->cfg.nnLambda; a.fac = (float)(1.0 / h->Bglobal);
  a.epsAnneal = h->cfg.epsAnneal; a.parity = parity; a.variant = h->dbgVariant;
  if (h->pushGrad) {      // (set by the step sequences around their weight-gradient launches)
    a.push.on = 1; a.push.nRanks = h->cfg.n_ranks; a.push.rank = h->cfg.rank; a.push.peers = h->xchg.dPeers;
    a.push.slotsOffset = h->xchg.slotsOffset; a.push.slotBytes = h->xchg.slotBytes; a.push.ctl = h->xchg.ctl; a.push.gBase = h->G;
    a.push.self = h->foldNow ? 1 : 0;      // (folded launch: launchWeightGrad)
  }
  return a;
}
SampleArgs sampleArgs(hl_learner* h, int parity, const long long* dFlat, bool computeEta) {
  SampleArgs sa{}; sa.sc = h->sc; sa.rp = h->rp; sa.bt = h->buf[parity].bt; sa.B = h->B; sa.dS = h->dS; sa.ldX0 = h->ldX0;
  sa.X0 = h->buf[parity].X0; sa.flatGiven = dFlat; sa.adamDraws = 1;      // thread 0's Saru seed: the only per-step draw from this generator (Optimizer.cpp:139, generators[thrID])
  // (recurrent layers read the raw states of their window straight from the replay: no gathered rows either)
  sa.parity = parity; sa.computeEta = computeEta ? 1 : 0; sa.backupRng = 0; sa.noGather = (h->preproc || h->recurrent) ? 1 : 0; sa.eta0 = (float)h->cfg.learnrate; sa.epsAnneal = h->cfg.epsAnneal;
  return sa;
}
// replica exchanges are part of the step: several replicas, or a communicator was attached to a
// single one (hl_comm_init with n_ranks == 1 runs the N > 1 sequence over a 1-rank RCCL communicator)
bool exchanging(const hl_learner* h) { return h->cfg.n_ranks > 1 || h->comm != nullptr; }
bool wired(const hl_learner* h) { return h->comm != nullptr || h->xchg.on; }      // the library itself exchanges (RCCL or xchg.hip)
PostArgs postArgs(hl_learner* h, int parity, int mode) {
  PostArgs pa{}; pa.sc = h->sc; pa.rp = h->rp; pa.bt = h->buf[parity].bt; pa.B = h->B; pa.mode = mode;
  pa.clipImpWeight = h->cfg.clipImpWeight; pa.epsAnneal = h->cfg.epsAnneal; pa.penalTol = h->cfg.penalTol;
  pa.maxObsGlobal = (double)h->maxObsGlobal; pa.batchGlobal = (double)h->Bglobal; pa.nRanks = exchanging(h) ? 2 : 1;   // > 1: use the exchanged counters
  pa.parity = parity; pa.eta0 = (float)h->cfg.learnrate; pa.aggStaged = h->fusedOk ? 1 : 0; pa.hasAdv = h->nAdv > 0 ? 1 : 0;
  pa.cntMsg = wired(h) ? h->G + h->nParams + CNT_MSG_OFFSET : nullptr;
  return pa;
}
HeadArgs headArgs(hl_learner* h, int parity) {
  const DevHidden& q = h->hid[h->nHidden - 1];
  HeadArgs ha{}; ha.sc = h->sc; ha.rp = h->rp; ha.bt = h->buf[parity].bt; ha.B = h->B; ha.dA = h->dA; ha.nDense = h->nDense; ha.nAdv = h->cfg.adv_kind == HL_ADV_GAUSSIAN ? h->nAdv : 0; ha.nOpt = h->nOpt; ha.nSig = h->nSig;
  ha.nOut = h->nOut; ha.H = q.size; ha.Yin = q.hasRes ? q.Rr : q.Y; ha.ldY = q.ldA; ha.Xlast = q.X; ha.Ylast = q.Y;
  ha.func = q.func; ha.params = h->W; ha.indWo = h->indWo; ha.indBo = h->indBo; ha.indBp = h->indBp; ha.ldWo = h->ldWo;
  ha.dOut = h->dOut; ha.ldDo = h->ldDo; ha.Dres = q.Dres; ha.D = q.D; ha.ldD = q.ldA; ha.parity = parity; ha.outFunc = h->cfg.nnOutputFunc;
  for (int i = 0; i < h->dA; ++i) ha.bounded[i] = h->cfg.bounded[i];
  return ha;
}

// prioritised samplers: Sampling::prepare() of the reference runs at the end of every step (updateSampler,
// MemoryProcessing.cpp:350) on the errors that step left behind -- here right before the draw, on the same data
int launchPerPrepare(hl_learner* h, hipStream_t s) {
  const long long n = std::max<long long>(h->nTransitions, (long long)h->order.size());
  if (n > h->perCap) {
    HIPCK(hipStreamSynchronize(s));
    for (void* q : {(void*)h->perProb, (void*)h->perKey, (void*)h->perKeyS, (void*)h->perCp, (void*)h->perIdx, (void*)h->perIdxS, h->perTemp, h->perScan}) if (q) hipFree(q);
    const size_t cap = (size_t)n + (size_t)n / 4 + 1024;
    HIPCK(devAlloc(&h->perProb, cap)); HIPCK(devAlloc(&h->perCp, cap));
    { h->perScanBytes = per_scan_scratch_bytes((long long)cap); unsigned char* t = nullptr; HIPCK(devAlloc(&t, h->perScanBytes)); h->perScan = t; }
    if (h->cfg.dataSamplingAlgo == HL_SAMPLE_PERRANK) {
      HIPCK(devAlloc(&h->perKey, cap)); HIPCK(devAlloc(&h->perKeyS, cap)); HIPCK(devAlloc(&h->perIdx, cap)); HIPCK(devAlloc(&h->perIdxS, cap));
      h->perTempBytes = per_sort_temp_bytes((long long)cap);
      unsigned char* t = nullptr; HIPCK(devAlloc(&t, h->perTempBytes)); h->perTemp = t;
    }
    h->perCap = (long long)cap;
  }
  PerArgs pa{}; pa.rp = h->rp; pa.nEpisodes = (int)h->order.size(); pa.algo = h->cfg.dataSamplingAlgo;
  pa.prob = h->perProb; pa.cp = h->perCp; pa.key = h->perKey; pa.keySorted = h->perKeyS; pa.idx = h->perIdx; pa.idxSorted = h->perIdxS;
  pa.temp = h->perTemp; pa.tempBytes = h->perTempBytes;
  pa.scan = h->perScan; pa.scanBytes = h->perScanBytes;
  HIPCK(timed(h, "per_prepare", s, [&] { return launch_per_prepare(pa, h->nTransitions, s); }));
  return HL_OK;
}
int launchSample(hl_learner* h, int parity, const long long* dFlat, bool computeEta, hipStream_t s) {
  SampleArgs sa = sampleArgs(h, parity, dFlat, computeEta);
  if (h->cfg.dataSamplingAlgo != HL_SAMPLE_UNIFORM && !dFlat) {
    int rc = launchPerPrepare(h, s); if (rc) return rc;
    sa.perAlgo = h->cfg.dataSamplingAlgo; sa.perCp = h->perCp;
    sa.perN = h->cfg.dataSamplingAlgo == HL_SAMPLE_PERSEQ ? (long long)h->order.size() : h->nTransitions;
  }
  HIPCK(timed(h, "step_tail_kernel", s, [&] { return launch_sample(sa, s); }));
  return HL_OK;
}
// `nextSample`: sampler phases of the NEXT step (buffer parity^1) ride along as extra workgroups:
// phase A with the first forward GEMM, phase B with the last one, phase C with the head kernel.
// `fusePost`: the bookkeeping of THIS step rides along the first backward launch.
ExtraArgs extraSample(hl_learner* h, int parityNext, int phases) {
  ExtraArgs ex{}; ex.role = 1; ex.phases = phases; ex.samp = sampleArgs(h, parityNext, nullptr, false);
  ex.samp.backupRng = 1;       // the minibatch drawn here may have to be discarded (dropPresample)
  return ex;
}
FusedArgs fusedArgs(hl_learner* h, int parity) {
  const DevHidden& d0 = h->hid[0]; const DevHidden& d1 = h->hid[1];
  FusedArgs fa{}; fa.sc = h->sc; fa.rp = h->rp; fa.bt = h->buf[parity].bt;
  fa.B = h->B; fa.dS = h->dS; fa.dA = h->dA; fa.nDense = h->nDense; fa.nOut = h->nOut; fa.H = d1.size; fa.parity = parity;
  fa.func = d1.func; fa.resN = d1.resW;
  fa.X0 = h->buf[parity].X0; fa.ldX0 = h->ldX0; fa.W = h->W;
  fa.indW0 = d0.indW; fa.indB0 = d0.indB; fa.indW1 = d1.indW; fa.indB1 = d1.indB; fa.indWr = d1.indWr; fa.indBr = d1.indBr;
  fa.indWo = h->indWo; fa.indBo = h->indBo; fa.indBp = h->indBp; fa.ldW0 = d0.ldW; fa.ldW1 = d1.ldW;
  fa.Y1 = d0.Y; fa.D1 = d0.D; fa.Dres1 = d0.Dres; fa.ldA0 = d0.ldA;
  fa.X2 = d1.X; fa.R2 = d1.Rr; fa.D2 = d1.D; fa.Dres2 = d1.Dres; fa.ldA1 = d1.ldA;
  fa.nLH = 2;
  if (h->fusedWideOk && h->nHidden == 3) {      // three equal hidden blocks (fusedw.hip)
    const DevHidden& d2 = h->hid[2];
    fa.nLH = 3; fa.indW2 = d2.indW; fa.indB2 = d2.indB; fa.indWr2 = d2.indWr; fa.indBr2 = d2.indBr; fa.ldW2 = d2.ldW; fa.resN2 = d2.resW;
    fa.X3 = d2.X; fa.R3 = d2.Rr; fa.D3 = d2.D; fa.Dres3 = d2.Dres; fa.ldA2 = d2.ldA;
  }
  fa.dOut = h->dOut; fa.ldDo = h->ldDo; fa.panelCtr = h->panelCtr; fa.variant = h->dbgVariant; fa.xcdSafe = h->xcdSafe ? 1 : 0;
  for (int i = 0; i < h->dA; ++i) if (h->cfg.bounded[i]) fa.boundedMask |= 1ull << i;
  return fa;
}
// forward + head + dX of the whole minibatch as one kernel (fused.hip); `nextSample`: sampler phases
// A and B of the NEXT step ride along
// `deferBeta`: the step before ran its bookkeeping with POST_DEFER: block 1 finishes it (farBetaPhase), the heads wait for beta
int launchFused(hl_learner* h, int parity, hipStream_t s, bool nextSample = false, bool deferBeta = false) {
  FusedArgs fa = fusedArgs(h, parity);
  ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
  // (draws, sort, redraws of the next minibatch here; its search and gather ride along the dW kernel: launchWeightGrad)
  if (nextSample) { ex = extraSample(h, parity ^ 1, PH_A | PH_B); pex = &ex; }
  if (deferBeta) { fa.deferBeta = 1; ex.post = postArgs(h, parity ^ 1, POST_BETA); pex = &ex; }
  HIPCK(timed(h, "fused_fwd_head_dx", s, [&] { return launch_fused(fa, h->Mmax, pex, s); }));
  return HL_OK;
}
HeadArgs headArgs(hl_learner* h, int parity);
// the wide variant (fusedw.hip): the same launch with the head's description next to the fused kernel's
int launchFusedWide(hl_learner* h, int parity, hipStream_t s, bool nextSample = false, bool deferBeta = false) {
  FusedArgs fa = fusedArgs(h, parity);
  const HeadArgs ha = headArgs(h, parity);
  ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
  if (nextSample) { ex = extraSample(h, parity ^ 1, PH_A | PH_B); pex = &ex; }
  if (deferBeta) { fa.deferBeta = 1; ex.post = postArgs(h, parity ^ 1, POST_BETA); pex = &ex; }
  HIPCK(timed(h, "fused_wide", s, [&] { return launch_fused_wide(fa, ha, h->Mmax, pex, s); }));
  return HL_OK;
}
int launchFront(hl_learner* h, int parity, hipStream_t s, bool gather);
// dense networks off the fused kernels: forward chain, head and input-gradient chain as one launch (gemm16.hip: step_chain_kernel)
// `wholeSampler`: the rider draws, sorts, searches and gathers the next minibatch (a few state components: no gather helpers needed)
// `deferBeta`: as launchFused
int launchStepChain(hl_learner* h, int parity, hipStream_t s, bool nextSample = false, bool wholeSampler = false, bool deferBeta = false) {
  const AdamHyper hyp = adamHyper(h, parity);
  const StepBuf& sb = h->buf[parity];
  { const int rc = launchFront(h, parity, s, true); if (rc) return rc; }
  HeadArgs ha = headArgs(h, parity);
  ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
  if (nextSample) { ex = extraSample(h, parity ^ 1, wholeSampler ? PH_ALL : (PH_A | PH_B)); pex = &ex; }
  if (deferBeta) { ha.deferBeta = 1; ex.post = postArgs(h, parity ^ 1, POST_BETA); pex = &ex; }
  int fIdx[HL_MAX_HIDDEN], xIdx[HL_MAX_HIDDEN];
  for (int j = 0; j < h->nHidden; ++j) fIdx[j] = sb.fwdIdx[j];
  const int nX = (int)sb.dxIdx.size();
  for (int i = 0; i < nX; ++i) xIdx[i] = sb.dxIdx[i];
  HIPCK(timed(h, "step_chain", s, [&] { return launch_step_chain(h->dProbs, fIdx, h->nHidden, xIdx, nX, h->chainHT, h->Mmax, h->panelCtr, ha, h->sc, hyp, pex, s); }));
  return HL_OK;
}
ConvArgs convArgs(hl_learner* h, int parity);
// (never inside a stream capture: the launch has to RUN before the flag is cleared)
int ensureConvPrep(hl_learner* h) {
  if (h->nConv == 0 || !h->convPrepStale) return HL_OK;
  const ConvArgs ca = convArgs(h, 0);
  HIPCK(launch_conv_prep(ca, h->stream));
  h->convPrepStale = false;
  return HL_OK;
}
ConvArgs convArgs(hl_learner* h, int parity) {
  ConvArgs ca{}; ca.sc = h->sc; ca.parity = parity; ca.B = h->convB; ca.nL = h->nConv;
  ca.W = h->W; ca.Wrw = h->W; ca.M1 = h->M1; ca.M2 = h->M2; ca.G = h->G;
  for (int l = 0; l < h->nConv; ++l) { ca.L[l] = h->cg[l]; ca.L[l].in = l == 0 ? h->buf[parity].X0 : h->cg[l - 1].Y; }
  return ca;
}
// training steps of a net whose first layer runs the row-block kernels: those read their input windows from the replay, the
// stacked rows X0 (28 KB per row at 84 x 84 x 4, written and read once per step) and their launch are not needed
bool convFromReplay(const hl_learner* h) {
  if (h->nConv == 0 || !h->cg[0].rbRows || h->noConvReplay || h->extras > 0) return false;
  const ConvGeo& g = h->cg[0];
  return (h->dS & 3) == 0 && (g.InX & 3) == 0 && g.InC % (1 + h->nApp) == 0 && (long long)(g.InC / (1 + h->nApp)) * g.InY * g.InX == h->dS;
}
void convSource(hl_learner* h, int parity, ConvArgs* ca) {
  const DevBatch& bt = h->buf[parity].bt;
  ca->src.on = 1; ca->src.S = h->rp.S; ca->src.mean = h->rp.stMean; ca->src.scale = h->rp.stScale; ca->src.dS = h->dS; ca->src.nApp = h->nApp;
  ca->src.slot = bt.slot; ca->src.t = bt.t; ca->src.nextSrc = bt.nextSrc;
  ca->L[0].rbKind = (h->convRowsAtari && conv_rows_atari(ca->L[0], ca->src)) ? 1 : 0;      // (the first layer of RACER_atari.json: geometry at compile time)
}
// `gather`: states with appended observations / convolutional input are assembled here, from the sampled slots
// (rollout inference writes the standardised rows itself)
// the launches in front of the first dense / recurrent layer: stacked rows, state variables beside the image, the conv stack
int launchFront(hl_learner* h, int parity, hipStream_t s, bool gather) {
  const StepBuf& sb = h->buf[parity];
  char nm[32];
  // convolutions in front of recurrent layers (training steps): rows = the samples' windows (rec.hip: window_rows_kernel)
  const bool windows = gather && h->recurrent && h->nConv > 0;
  DevBatch bt = sb.bt; DevScalars* sc = h->sc;
  if (windows) {
    WinRowsArgs wa{}; wa.sc = h->sc; wa.scW = h->scW; wa.slot = bt.slot; wa.t = bt.t; wa.nextSrc = bt.nextSrc; wa.B = h->B; wa.K = h->recK; wa.nBPTT = h->recWin - 1;
    wa.parity = parity; wa.slotW = h->winSlot; wa.tW = h->winT; wa.nextSrcW = h->winNextSrc;
    HIPCK(timed(h, "window_rows", s, [&] { return launch_window_rows(wa, s); }));
    bt.slot = h->winSlot; bt.t = h->winT; bt.nextSrc = h->winNextSrc; sc = h->scW;
  }
  const bool fromReplay = gather && convFromReplay(h);
  if (h->preproc && gather && !fromReplay) {
    StackGatherArgs ga{}; ga.sc = sc; ga.rp = h->rp; ga.bt = bt; ga.B = h->convB; ga.dS = h->dS; ga.nApp = h->nApp;
    ga.parity = parity; ga.X0 = sb.X0; ga.ldX0 = h->ldX0;
    HIPCK(timed(h, "stack_gather", s, [&] { return launch_stack_gather(ga, h->convMmax, s); }));
  }
  if (h->extras > 0)
    HIPCK(timed(h, "extras_copy", s, [&] { return launch_extras_copy(sc, parity, sb.X0, h->ldX0, h->dIn - h->extras, h->extras, h->hid[0].Y, h->hid[0].ldA, h->convMmax, s); }));
  if (h->nConv > 0) {
    ConvArgs ca = convArgs(h, parity);
    ca.sc = sc;
    if (fromReplay) { convSource(h, parity, &ca); ca.src.slot = bt.slot; ca.src.t = bt.t; ca.src.nextSrc = bt.nextSrc; }
    // filters -> the kernels' LDS layouts: kept current by the Adam pass of the filter gradients (conv_reduce_adam_kernel);
    // rebuilt here only after something else wrote the weights (start-up, hl_set_params, a restart) or where Adam runs
    // elsewhere (replica exchange)
    if (exchanging(h)) HIPCK(timed(h, "conv_prep", s, [&] { return launch_conv_prep(ca, s); }));
    else if (h->convPrepStale) return fail(h, HL_ERR_STATE, "convolution filter layouts are stale (ensureConvPrep was not called)");
    for (int l = 0; l < h->nConv; ++l) {
      snprintf(nm, sizeof(nm), "conv_fwd%d", l);
      if (l == 1 && h->convTail.atari) {      // layers 1 .. 3 of the RACER_atari stack: one launch, a workgroup per row (convt.hip)
        HIPCK(timed(h, "conv_fwd_tail", s, [&] { return launch_conv_fwd_tail(ca, h->convTail, h->convMmax, s); }));
        break;
      }
      if (ca.L[l].rbRows) HIPCK(timed(h, nm, s, [&] { return launch_conv_forward_rows(ca, l, h->convMmax, s); }));
      else
      HIPCK(timed(h, nm, s, [&] { return launch_conv_forward(ca, l, h->convMmax, s); }));
    }
  }
  return HL_OK;
}
int launchForward(hl_learner* h, int parity, hipStream_t s, bool nextSample = false, bool gather = true) {
  const AdamHyper hyp = adamHyper(h, parity);
  const StepBuf& sb = h->buf[parity];
  char nm[32];
  { const int rc = launchFront(h, parity, s, gather); if (rc) return rc; }
  const int j0 = h->nConv > 0 ? 1 : 0;
  if (h->chainOk) {      // every dense layer in one launch, sampler phases A and B of the next step riding along
    ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
    if (nextSample) { ex = extraSample(h, parity ^ 1, PH_A | PH_B); pex = &ex; }
    int idx[HL_MAX_HIDDEN];
    for (int j = j0; j < h->nHidden; ++j) idx[j - j0] = sb.fwdIdx[j];
    HIPCK(timed(h, "fwd_chain", s, [&] { return launch_fwd_chain(h->dProbs, idx, h->nHidden - j0, h->chainHT, h->Mmax, h->panelCtr, h->sc, hyp, pex, nullptr, s); }));
    return HL_OK;
  }
  for (int j = j0; j < h->nHidden; ++j) {
    snprintf(nm, sizeof(nm), "gemm16_fwd%d", j);
    ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
    if (nextSample) {
      int ph = 0;
      if (j == j0) ph |= PH_A;
      if (j == h->nHidden - 1) ph |= PH_B;
      if (ph) { ex = extraSample(h, parity ^ 1, ph); pex = &ex; }
    }
    const int role = j == j0 ? GEMM_ROLE_FWD0 : GEMM_ROLE_FWD;
    if ((h->bigMm & 4) && big_mm_ok(h->hostProbs[sb.fwdIdx[j]]))
      HIPCK(timed(h, nm, s, [&] { return launch_big_mm(h->hostProbs[sb.fwdIdx[j]], h->sc, parity, s); }));
    else
    if ((h->bigMm & 1) && big_panel_ok(h->hostProbs[sb.fwdIdx[j]]))
      HIPCK(timed(h, nm, s, [&] { return launch_big_panel(h->hostProbs[sb.fwdIdx[j]], h->sc, parity, s); }));
    else
    if (gemm_oneshot_ok(GEMM_F, h->hid[j].nIn))      // long reductions (the dense layer behind a convolution stack): every operand load in flight at once
      HIPCK(timed(h, nm, s, [&] { return launch_gemm_oneshot(role, h->dProbs + sb.fwdIdx[j], h->hid[j].nIn, sb.fwdBlocks[j], h->sc, hyp, pex, s); }));
    else
    HIPCK(timed(h, nm, s, [&] { return launch_gemm(role, h->dProbs + sb.fwdIdx[j], 1, sb.fwdBlocks[j], h->sc, hyp, pex, s); }));
  }
  return HL_OK;
}
int launchHead(hl_learner* h, int parity, hipStream_t s, bool nextSample = false) {
  const HeadArgs ha = headArgs(h, parity);
  ExtraArgs ex{}; const ExtraArgs* pex = nullptr;
  // (recurrent nets have no forward GEMM launches: the whole sampler of the next step rides along the head kernel)
  if (nextSample) {
    // recurrent nets (no forward GEMM launches): draws and sort of the next minibatch ride here, its index search the weight-gradient
    // launch (launchBackward, sampleC) -- the whole sampler (12 us) was this launch's longest workgroup
    ex = extraSample(h, parity ^ 1, h->recurrent ? (PH_A | PH_B) : PH_C); pex = &ex;
    if (!ex.samp.noGather) {      // the gather of 2 B rows of dS floats: ~1024 floats per helper workgroup, at most 31 of them
      const long long fl = 2LL * h->B * h->dS;
      ex.helpers = (int)std::min<long long>(31, fl / 1024);
      if (ex.helpers > 0) ex.phases |= PH_PUBLISH;
      // (the sorted indices of a dense net's next minibatch were left by phase B in an earlier launch: the helpers search them themselves)
      if (ex.helpers > 0 && !h->recurrent) ex.samp.selfSearch = 1;
    }
  }
  if (h->panelHead && panel_head_ok(ha)) { HIPCK(timed(h, "panel_head", s, [&] { return launch_panel_head(ha, h->Mmax, pex, s); })); return HL_OK; }
  HIPCK(timed(h, "head_kernel", s, [&] { return launch_head(ha, h->Mmax, pex, s); }));
  return HL_OK;
}
// fuseAdam: apply the Adam update inside the dW epilogue (only valid without a gradient exchange)
// dW launch of the fused path: the dX contractions were done by the fused kernel; riders: the
// bookkeeping of THIS step and sampler phase C of the NEXT one
// `fold` (replicas over peer windows, replayed steps): the gradient's exchange, Adam and the step's closing bookkeeping run inside this
// launch (gemm16.hip: dw_table_kernel, xchg_dev.h) -- the caller issues no exchange launch; only with h->foldOk, the bookkeeping rider
// and a problem table that travels in the kernel arguments
bool foldUsable(const hl_learner* h, int parity) {
  return h->foldOk && h->pushGrad && h->xchg.on && h->buf[parity].dwCount <= DW_TABLE_MAX && ((h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS) & 3) == 0;
}
int launchWeightGrad(hl_learner* h, int parity, bool fuseAdam, hipStream_t s, bool fusePost, bool nextSampleC,
                     int postMode = POST_AGG | POST_BETA, bool fold = false) {
  struct FoldScope { hl_learner* h; ~FoldScope() { h->foldNow = false; } } foldScope{h};
  if (fold && !(fusePost && !fuseAdam && foldUsable(h, parity))) return fail(h, HL_ERR_STATE, "folded weight-gradient launch asked for where it cannot run");
  h->foldNow = fold;
  const AdamHyper hyp = adamHyper(h, parity);
  const StepBuf& sb = h->buf[parity];
  ExtraArgs exP{}, exC{};
  if (fusePost) { exP.role = 2; exP.post = postArgs(h, parity, postMode); }
  if (nextSampleC) exC = extraSample(h, parity ^ 1, PH_C);
  if (sb.dwCount <= DW_TABLE_MAX) {   // problem table in the kernel arguments
    const DwTable& tbl = fuseAdam ? sb.dwTableAdam : sb.dwTable;
    if (nextSampleC && !exC.samp.noGather) { exC.phases |= PH_PUBLISH; exC.helpers = 7; exC.samp.tagSeq = 1; exC.samp.selfSearch = 1; }
    FoldArgs fo{};
    if (fold) {
      const long long n = (long long)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS, bytes = n * 4;
      fo.on = 1; fo.nCh = xchg_chunks(bytes, h->xchg.maxChunks);      // (as launch_xchg_allreduce cuts the message)
      fo.nTiles = sb.dwBlocks; fo.msg = h->G; fo.n = n; fo.W = h->W; fo.M1 = h->M1; fo.M2 = h->M2; fo.nAdam = h->nParams;
      fo.ctl = h->xchg.ctl; fo.timeoutTicks = h->xchgTimeoutTicks;
      if ((size_t)bytes > h->xchg.slotBytes) return fail(h, HL_ERR_COMM, "exchange message larger than the window slot");
    }
    HIPCK(timed(h, fold ? "dw_table_fold" : "dw_table_kernel", s, [&] { return launch_dw_table(tbl, sb.dwBlocks, h->sc, hyp, fusePost ? &exP : nullptr, s, nextSampleC ? &exC : nullptr, fold ? &fo : nullptr); }));
    if (fold) h->nCollectives += 1;
    return HL_OK;
  }
  // (more problems than the argument table holds: the table in device memory; the same riders, the gather helpers behind them)
  if (nextSampleC && !exC.samp.noGather) { exC.phases |= PH_PUBLISH; exC.helpers = 7; exC.samp.tagSeq = 1; exC.samp.selfSearch = 1; }
  HIPCK(timed(h, "gemm16_dw", s, [&] {
    return launch_gemm(GEMM_ROLE_DW, h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx), sb.dwCount, sb.dwBlocks, h->sc, hyp,
                       nextSampleC ? &exC : nullptr, s, fusePost ? &exP : nullptr); }));
  return HL_OK;
}
// `sampleC`: the index search of the NEXT minibatch (sampler phase C; recurrent nets) rides the weight-gradient launch
// `skipDx`: the input gradients were taken by the launch in front (gemm16.hip: step_chain_kernel): the weight gradients and their riders only
int launchBackward(hl_learner* h, int parity, bool fuseAdam, hipStream_t s, bool fusePost = false, int postMode = POST_AGG | POST_BETA, bool sampleC = false, bool skipDx = false) {
  const AdamHyper hyp = adamHyper(h, parity);
  const StepBuf& sb = h->buf[parity];
  ExtraArgs ex{}, exF{}; const ExtraArgs* pex = nullptr; const ExtraArgs* pexF = nullptr;
  const bool noDx = skipDx || sb.dxIdx.empty();
  // one replica, bookkeeping riding the first dX launch: its far-policy count and the beta update move on to the dW launch (the next
  // reader of beta is the head kernel of the step after), off what was that launch's longest workgroup
  // (no dX launch -- recurrent layers, a single hidden layer --: the bookkeeping rides the dW launch; where a split-row join follows,
  // count and beta ride that one)
  // recurrent nets: the weight gradients over all (sample, step) rows as ONE launch (no row chunks, no join) unless this replica
  // pushes its tiles into peer windows from the launch
  const bool wideDw = h->wideDw && sb.dwWideIdx >= 0 && !hyp.push.on && sb.bigDw.empty();
  const bool viaSplit = noDx && sb.splitMaxMN > 0 && !wideDw;
  PostArgs fbSplit{};
  if (fusePost && postMode == (POST_AGG | POST_BETA) && (!noDx || viaSplit) && !h->noDeferBeta) {
    postMode |= POST_DEFER;
    if (viaSplit) fbSplit = postArgs(h, parity, POST_BETA);
    else { exF.role = 3; exF.post = postArgs(h, parity, POST_BETA); pexF = &exF; }
  }
  const bool splitRider = viaSplit && (postMode & POST_DEFER);
  if (fusePost) { ex.role = 2; ex.post = postArgs(h, parity, postMode); pex = &ex; }
  char nm[32];
  for (size_t i = 0; i < sb.dxIdx.size() && !skipDx; ++i) {
    snprintf(nm, sizeof(nm), "gemm16_dx%d", h->nHidden - 1 - (int)i);
    const int jx = h->recurrent ? 1 : h->nHidden - 1 - (int)i;          // problem i back-propagates through block jx: reduction over its outputs
    const int Kx = h->recurrent ? std::max(h->hid[jx].lstm, 1) * h->hid[jx].size : h->hid[jx].size;      // (recurrent layer under a conv stack: over its gates)
    if ((h->bigMm & 4) && !pex && big_mm_ok(h->hostProbs[sb.dxIdx[i]]))
      HIPCK(timed(h, nm, s, [&] { return launch_big_mm(h->hostProbs[sb.dxIdx[i]], h->sc, parity, s); }));
    else
    if ((h->bigMm & 1) && !pex && big_panel_ok(h->hostProbs[sb.dxIdx[i]]))
      HIPCK(timed(h, nm, s, [&] { return launch_big_panel(h->hostProbs[sb.dxIdx[i]], h->sc, parity, s); }));
    else
    if (gemm_oneshot_ok(GEMM_X, Kx))
      HIPCK(timed(h, nm, s, [&] { return launch_gemm_oneshot(GEMM_ROLE_DX, h->dProbs + sb.dxIdx[i], Kx, sb.dxBlocks[i], h->sc, hyp, i == 0 ? pex : nullptr, s); }));
    else
    HIPCK(timed(h, nm, s, [&] { return launch_gemm(GEMM_ROLE_DX, h->dProbs + sb.dxIdx[i], 1, sb.dxBlocks[i], h->sc, hyp, i == 0 ? pex : nullptr, s); }));
  }
  bool denseMerged = false;
  if (h->nConv > 0) {   // convolutional layers: input gradients from the last one down, then every filter gradient (+ Adam)
    ConvArgs ca = convArgs(h, parity);
    if (convFromReplay(h)) {      // (the backward pass belongs to a training step: the rows were never stacked)
      convSource(h, parity, &ca);
      if (h->recurrent) { ca.src.slot = h->winSlot; ca.src.t = h->winT; ca.src.nextSrc = h->winNextSrc; }      // (windows: launchFront)
    }
    if (h->recurrent) ca.sc = h->scW;
    int nRb = 0, lRb = -1;
    for (int l = 0; l < h->nConv; ++l) if (ca.L[l].rbRows) { ++nRb; lRb = l; }
    // the dense layers' weight-gradient tiles depend on the deltas only, like the filter gradients: with one workgroup per tile and
    // no second rider they join the filter-gradient launch (their Adam pass touches no parameter conv_reduce_adam touches)
    denseMerged = nRb == 1 && h->convDwBlocks > 0 && h->convDwDense && h->directDw && !h->recurrent && sb.splitMaxMN == 0 && sb.bigDw.empty()
                  && !hyp.push.on && sb.dwBlocks >= h->directDwMinTiles && !sampleC && !(noDx && pex);
    for (int i = 0; denseMerged && i < sb.dwCount; ++i) if (h->hostProbs[sb.dwIdx + i].K > 128) denseMerged = false;      // (the instantiated row batches)
    // ... and the tiles behind the convolution biases' column sums (the table's first nConv problems) need no delta of a
    // convolutional layer: they ride the input-gradient launches of the unstrided layers (a few hundred workgroups each), in equal shares
    const GemmProblem* dwProbs = h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx);
    int rideT0 = sb.dwBlocks, nRideL = 0;
    if (denseMerged && h->convDxRide && sb.dwCount > h->nConv) {
      for (int l = h->nConv - 1; l >= 1; --l) if (conv_dx_rides(ca.L[l])) ++nRideL;
      if (nRideL) rideT0 = h->hostProbs[sb.dwIdx + h->nConv].tileStart;
    }
    int rideNext = rideT0, rideLeft = nRideL;
    if (h->convTail.on) {      // the input gradients of every layer behind the first: one launch, a workgroup per row (convt.hip)
      rideT0 = sb.dwBlocks;
      HIPCK(timed(h, "conv_back", s, [&] { return launch_conv_back(ca, h->convTail, s); }));
    } else
    for (int l = h->nConv - 1; l >= 1; --l) {
      snprintf(nm, sizeof(nm), "conv_dx%d", l);
      DenseRide rd{}; const DenseRide* prd = nullptr;
      if (rideLeft > 0 && conv_dx_rides(ca.L[l])) {
        const int share = (sb.dwBlocks - rideNext + rideLeft - 1) / rideLeft;
        rd.probs = dwProbs; rd.nProbs = sb.dwCount; rd.tile0 = rideNext; rd.tile1 = rideNext + share; rd.hyp = hyp; prd = &rd;
        rideNext += share; --rideLeft;
      }
      HIPCK(timed(h, nm, s, [&] { return launch_conv_dx(ca, l, s, prd); }));
    }
    if (denseMerged) {
      HIPCK(timed(h, "conv_dw_dense", s, [&] {
        return launch_conv_dw_dense(ca, lRb, h->convDwBlocks, dwProbs, sb.dwCount, rideT0, hyp, pexF, s); }));
    } else
    if (nRb == 1 && h->convDwBlocks > 0) {      // the two filter-gradient launches depend on the deltas only: one launch
      HIPCK(timed(h, "conv_dw_all", s, [&] { return launch_conv_dw_all(ca, lRb, h->convDwBlocks, s); }));
    } else {
    for (int l = 0; l < h->nConv; ++l) if (ca.L[l].rbRows) HIPCK(timed(h, "conv_dw_rows", s, [&] { return launch_conv_dw_rows(ca, l, s); }));
    if (h->convDwBlocks > 0) HIPCK(timed(h, "conv_dw", s, [&] { return launch_conv_dw(ca, h->convDwBlocks, s); }));
    }
    ca.sc = h->sc;      // (the learning rate of the step)
    HIPCK(timed(h, "conv_reduce_adam", s, [&] { return launch_conv_reduce_adam(ca, hyp, fuseAdam ? 1 : 0, s); }));
    if (denseMerged) return HL_OK;
  }
  // a single hidden layer has no dX launch: the bookkeeping then rides along the dW launch.  It
  // writes etaEff[parity^1] only, never the slot the fused Adam of this launch reads.
  const ExtraArgs* pexW = noDx ? pex : nullptr;
  ExtraArgs exC{};
  if (sampleC && pexF) return fail(h, HL_ERR_STATE, "no rider slot left for the sampler's index search on the weight-gradient launch");
  if (sampleC) { exC = extraSample(h, parity ^ 1, PH_C); pexF = &exC; }
  // large batches: the dense layers' weight gradients as 64 x 64 tiles over row chunks (bigmm.hip); joined by splitk_reduce below
  if (!sb.bigDw.empty()) {
    BigDwList L{}; L.n = (int)sb.bigDw.size();
    for (int i = 0; i < L.n; ++i) { L.idx[i] = sb.bigDw[i] - sb.dwIdx; L.start[i + 1] = L.start[i] + big_dw_blocks(h->hostProbs[sb.bigDw[i]]); }
    HIPCK(timed(h, "big_dw", s, [&] { return launch_big_dw(h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx), L, s); }));
  }
  if (wideDw) {
    HIPCK(timed(h, "dw_wide", s, [&] {
      return launch_dw_wide(h->dProbs + (fuseAdam ? sb.dwWideAdamIdx : sb.dwWideIdx), sb.dwCount, sb.dwWideBlocks, DW_WIDE_Q, h->widePart, h->wideCtr, h->sc, hyp, pexW, pexF, s); }));
    return HL_OK;
  }
  // minibatch rows only and many tiles (the dense layer behind a convolutional stack: 1630): operands straight from memory into the
  // MFMA, no staging and no barrier in front of the reduction (dw_wide_kernel with one workgroup per tile) -- shorter workgroups
  if (h->directDw && sb.splitMaxMN == 0 && sb.bigDw.empty() && !hyp.push.on && sb.dwBlocks >= h->directDwMinTiles) {
    HIPCK(timed(h, "dw_direct", s, [&] {
      return launch_dw_wide(h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx), sb.dwCount, sb.dwBlocks, 1, nullptr, nullptr, h->sc, hyp, pexW, pexF, s); }));
    return HL_OK;
  }
  HIPCK(timed(h, "gemm16_dw", s, [&] {
    return launch_gemm(GEMM_ROLE_DW, h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx), sb.dwCount, sb.dwBlocks, h->sc, hyp, pexW, s, pexF); }));
  if (sb.splitMaxMN > 0)
    HIPCK(timed(h, "splitk_reduce", s, [&] {
      return launch_splitk_reduce(h->dProbs + (fuseAdam ? sb.dwAdamIdx : sb.dwIdx), sb.dwCount, sb.splitMaxMN, h->sc, hyp, s, splitRider ? &fbSplit : nullptr); }));
  return HL_OK;
}
int launchAdam(hl_learner* h, int parity) {
  AdamArgs aa{}; aa.sc = h->sc; aa.W = h->W; aa.M1 = h->M1; aa.M2 = h->M2; aa.G = h->G; aa.n = h->nParams;
  aa.eta0 = (float)h->cfg.learnrate; aa.lambda = (float)h->cfg.nnLambda; aa.fac = (float)(1.0 / h->Bglobal);
  aa.epsAnneal = h->cfg.epsAnneal; aa.parity = parity;
  HIPCK(timed(h, "adam_kernel", h->stream, [&] { return launch_adam(aa, h->stream); }));
  return HL_OK;
}
int launchPost(hl_learner* h, int parity, int mode, hipStream_t s) {
  PostArgs pa = postArgs(h, parity, mode);
  if (h->bigBatch && (mode & POST_AGG)) {      // large batches: the episode records by one workgroup per 256 samples
    HIPCK(timed(h, "post_agg_chunks", s, [&] { return launch_post_agg_chunks(pa, s); }));
    pa.aggChunk = 2;
  }
  HIPCK(timed(h, "post_kernel", s, [&] { return launch_post(pa, s); }));
  return HL_OK;
}

// every 1000th step: Episode::updateCumulative + full Retrace sweep, then reward/state statistics
int launchPeriodicSweep(hl_learner* h) { return runSweep(h, nullptr, (int)h->order.size(), 1); }
int launchMoments(hl_learner* h) {
  const int nb = moments_blocks((int)h->order.size());
  if (nb > h->momBlocksCap) {
    HIPCK(devGrow(&h->dMomPartial, 0, (size_t)nb * 2 * (h->dS + 1), h->stream)); h->momBlocksCap = nb;
  }
  MomentsArgs ma{}; ma.sc = h->sc; ma.rp = h->rp; ma.dS = h->dS; ma.nEpisodes = (int)h->order.size();
  ma.partial = h->dMomPartial; ma.nBlocks = nb; ma.moments = h->dMoments;
  HIPCK(timed(h, "moments_kernel", h->stream, [&] { return launch_moments(ma, h->stream); }));
  return HL_OK;
}
int launchMomentsApply(hl_learner* h, bool bInit, double rRateFac) {
  MomentsArgs ma{}; ma.sc = h->sc; ma.rp = h->rp; ma.dS = h->dS; ma.moments = h->dMoments;
  ma.bInit = bInit ? 1 : 0; ma.learnrate = h->cfg.learnrate; ma.epsAnneal = h->cfg.epsAnneal; ma.rRateFac = rRateFac;
  HIPCK(launch_moments_apply(ma, h->stream));
  return HL_OK;
}

// MemoryProcessing::applyEpisodesRemovalAlgo (MemoryProcessing.cpp:327-351): host bookkeeping + device far-policy count.
// "oldest": the back of the order.  Other rules of getERfilterAlgo (:261-298): the episode the reference's comparator puts
// last -- largest far-policy fraction / largest average D_KL / smallest average squared error --, the older one among equal
// keys; the keys are the per-episode aggregates the device maintains, fetched when a removal may be due.
bool evictionDue(const hl_learner* h);
int prepareExact(hl_learner* h, int n);
bool graphUsable(const hl_learner* h, int U, int p0);
int touchReplay(hl_learner* h);
int ensureConvPrep(hl_learner* h);
int applyRemoval(hl_learner* h) {
  bool any = false;
  const int filter = h->cfg.ERoldSeqFilter;
  if (evictionDue(h)) { int rc = refreshInsertionStats(h); if (rc) return rc; }     // the statistics pass precedes the removals
  if (filter == HL_ER_OLDEST) {
    while (!h->order.empty() && h->nTransitions - (long long)h->order.back().N > h->maxObsLocal) {
      const EpMeta e = h->order.back();
      h->nTransitions -= e.N - 1; h->freeEids.push_back(e.eid); h->order.pop_back(); any = true; h->minLenAtN = -1;
    }
  } else if (evictionDue(h)) {
    std::vector<float> agg((size_t)h->nextEid * AGG_N);
    HIPCK(hipMemcpyAsync(agg.data(), h->rp.epAgg, agg.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    const int col = filter == HL_ER_FARPOLFRAC ? AGG_FRACFAR : (filter == HL_ER_MAXKLDIV ? AGG_AVGKL : AGG_AVGSQERR);
    const float sgn = filter == HL_ER_MINERROR ? -1.f : 1.f;
    while (!h->order.empty()) {
      size_t v = h->order.size() - 1;
      float best = sgn * agg[(size_t)h->order[v].eid * AGG_N + col];
      for (size_t i = h->order.size() - 1; i-- > 0;) {             // from the oldest towards the newest: ties keep the older one
        const float k = sgn * agg[(size_t)h->order[i].eid * AGG_N + col];
        if (k > best) { best = k; v = i; }
      }
      const EpMeta e = h->order[v];
      if (h->nTransitions - (long long)e.N <= h->maxObsLocal) break;
      h->nTransitions -= e.N - 1; h->freeEids.push_back(e.eid); h->order.erase(h->order.begin() + (long)v); any = true; h->minLenAtN = -1;
    }
  }
  if (any) { h->tableDirty = true; h->countsDirty = true; }
  return HL_OK;
}

// the message also carries the parameter tail up to the counter chunks (postPart, cntMsg): one collective per step
// peer windows (hl_xchg_connect) take precedence over the RCCL communicator
int xchgAllreduce(hl_learner* h, void* buf, size_t n, int dtype, int fuseParity = -1) {
  XchgArgs xa{};
  if (fuseParity >= 0) {        // the gradient message of a step: Adam and the counters' bookkeeping inside the same kernel
    xa.fuse = 1;
    xa.adam.sc = h->sc; xa.adam.W = h->W; xa.adam.M1 = h->M1; xa.adam.M2 = h->M2; xa.adam.G = h->G; xa.adam.n = h->nParams;
    xa.adam.eta0 = (float)h->cfg.learnrate; xa.adam.lambda = (float)h->cfg.nnLambda; xa.adam.fac = (float)(1.0 / h->Bglobal);
    xa.adam.epsAnneal = h->cfg.epsAnneal; xa.adam.parity = fuseParity;
    xa.post = postArgs(h, fuseParity, POST_BETA);
    xa.pushed = h->pushGrad ? h->nParams : 0;      // (the launch that produced this gradient pushed it)
  } xa.msg = buf; xa.n = (long long)n; xa.nRanks = h->cfg.n_ranks; xa.rank = h->cfg.rank; xa.peers = h->xchg.dPeers;
  xa.slotsOffset = h->xchg.slotsOffset; xa.slotBytes = h->xchg.slotBytes; xa.ctl = h->xchg.ctl; xa.sc = h->sc;
  xa.timeoutTicks = h->xchgTimeoutTicks; xa.maxChunks = h->xchg.maxChunks;
  if (n * (dtype == 0 ? 4 : 8) > h->xchg.slotBytes) return fail(h, HL_ERR_COMM, "exchange message larger than the window slot");
  HIPCK(timed(h, "xchg_allreduce", h->stream, [&] { return launch_xchg_allreduce(xa, dtype, h->stream); }));
  h->nCollectives += 1;
  // anything but a step's gradient (start-up counters, moments, the weight broadcast) leaves its bytes in the slots: zeroed again, so
  // that the gradient slots hold zeros wherever no tile of a pushing launch writes
  if (fuseParity < 0 && h->pushOk) HIPCK(launch_xchg_clean(h->xchg.win, h->xchg.slotsOffset, h->xchg.slotBytes, h->cfg.n_ranks, h->xchg.ctl, (long long)(n * (dtype == 0 ? 4 : 8)), h->stream));
  return HL_OK;
}
int allreduceGrad(hl_learner* h) {
  if (!exchanging(h)) return HL_OK;
  const size_t n = (size_t)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS;
  if (h->xchg.on) return xchgAllreduce(h, h->G, n, 0);
  if (!h->comm) return fail(h, HL_ERR_COMM, "n_ranks > 1 but neither hl_xchg_connect nor hl_comm_init was called");
  NCCLCK(ncclAllReduce(h->G, h->G, n, ncclFloat, ncclSum, h->comm, h->stream));
  h->nCollectives += 1;
  return HL_OK;
}
int allreduceCounters(hl_learner* h) {      // start-up only (hl_initialize); steps carry the counters in the gradient message
  if (h->xchg.on) return xchgAllreduce(h, h->sc->cnt, 4, 2);
  if (!h->comm) return HL_OK;
  NCCLCK(ncclAllReduce(h->sc->cnt, h->sc->cnt, 4, ncclInt64, ncclSum, h->comm, h->stream));
  h->nCollectives += 1;
  return HL_OK;
}
int allreduceMoments(hl_learner* h) {
  if (h->xchg.on) return xchgAllreduce(h, h->dMoments, (size_t)(2 * h->dS + 3), 1);
  if (!h->comm) return HL_OK;
  NCCLCK(ncclAllReduce(h->dMoments, h->dMoments, (size_t)(2 * h->dS + 3), ncclDouble, ncclSum, h->comm, h->stream));
  h->nCollectives += 1;
  return HL_OK;
}

// (rules other than "oldest": whether the episode to go can go is only known once its key has been fetched; over budget by
// more than the shortest possible episode is the necessary condition)
bool evictionDue(const hl_learner* h) {
  if (h->order.empty()) return false;
  if (h->cfg.ERoldSeqFilter != HL_ER_OLDEST) {
    // which episode the rule picks depends on aggregates that live on the device, but no pick can leave unless even the shortest
    // stored episode could: without this bound a full replay sat "due" for good and every step took the eager route with a
    // device-to-host copy of the aggregates
    // (invalidated -- minLenAtN = -1 -- wherever the order changes: append, removal, restart; the two counters alone can stay
    //  equal across a removal plus arrivals that lower the minimum)
    if (h->minLenAtN != h->nTransitions || h->minLenAtCount != h->order.size()) {
      int minN = INT_MAX;
      for (const EpMeta& e : h->order) minN = std::min(minN, e.N);
      h->minLen = minN; h->minLenAtN = h->nTransitions; h->minLenAtCount = h->order.size();
    }
    return h->nTransitions - (long long)h->minLen > h->maxObsLocal;
  }
  return h->nTransitions - (long long)h->order.back().N > h->maxObsLocal;
}

// seg: -1 = the whole recurrent stack (one layer type); 0 / 1 = lower ("RNN" encoder layers) / upper segment of a two-type stack
RecArgs recArgs(hl_learner* h, int parity, int seg = -1) {
  const DevHidden& q = h->hid[h->nHidden - 1];
  RecArgs ra{}; ra.sc = h->sc; ra.rp = h->rp; ra.bt = h->buf[parity].bt; ra.B = h->B; ra.dS = h->dS;
  const int j0 = h->nConv > 0 ? 1 : 0;      // (hid[0] of a convolutional net is its last convolution: its rows are the first layer's input)
  const int jBeg = seg == 1 ? j0 + h->recSplit : j0, jEnd = seg == 0 ? j0 + h->recSplit : h->nHidden;
  ra.nL = jEnd - jBeg;
  ra.K = h->recK; ra.nBPTT = h->recWin - 1; ra.W = h->W; ra.gates = h->hid[jBeg].lstm; ra.func = h->cfg.nnFunc; ra.nApp = (j0 || seg == 1) ? 0 : h->nApp;
  for (int j = jBeg; j < jEnd; ++j) ra.L[j - jBeg] = h->rec[j];
  if (h->recTm && seg < 0) { ra.tmT = h->tmT; ra.tmSteps = h->tmSteps; ra.tmNext = h->tmNext; for (int j = jBeg; j < jEnd; ++j) { ra.tmER[j - jBeg] = h->tmER[j]; ra.tmSD[j - jBeg] = h->tmSD[j]; ra.tmFP[j - jBeg] = h->tmFP[j]; ra.tmCtrOff[j - jBeg] = h->tmCtrOff[j]; ra.tmET[j - jBeg] = h->tmET[j]; } ra.tmCtr = h->tmCtr; ra.tmCtrN = h->tmCtrN; }
  if (seg == 1) { ra.Xin = h->segY; ra.ldXin = h->ldSeg; }
  else if (j0) { ra.Xin = h->hid[0].Y; ra.ldXin = h->hid[0].ldA; }
  if (seg == 0) { ra.YoutRows = h->segY; ra.ldYR = h->ldSeg; ra.DresRows = h->segDres; ra.ldDR = h->ldSeg; }
  else { ra.Yout = q.hasRes ? q.Rr : q.Y; ra.ldY = q.ldA; ra.Dres = q.Dres; ra.ldD = q.ldA; }
  return ra;
}
// forward, head, backward (dX and dW) of buffer `parity`, eager, no riders
int launchMlp(hl_learner* h, int parity, bool fuseAdam, hipStream_t s) {
  if (h->fusedOk) {
    int rc = launchFused(h, parity, s); if (rc) return rc;
    return launchWeightGrad(h, parity, fuseAdam, s, false, false);
  }
  if (h->recurrent) {      // LSTM layers: window forward, head, back-propagation through time, then the common dW (+ Adam) launch
    if (h->nConv > 0) { const int rc = launchFront(h, parity, s, true); if (rc) return rc; }      // the windows' rows through the conv stack
    if (h->recSplit) {      // two layer types: the "RNN" encoder layers, then the MGU layers on their rows; backward the other way round
      const RecArgs lo = recArgs(h, parity, 0), up = recArgs(h, parity, 1);
      const StepBuf& sb = h->buf[parity];
      HIPCK(timed(h, "rec_forward_lower", s, [&] { return launch_rec_forward(lo, s); }));
      HIPCK(timed(h, "rec_forward", s, [&] { return launch_rec_forward(up, s); }));
      int rc = launchHead(h, parity, s); if (rc) return rc;
      HIPCK(timed(h, "rec_backward", s, [&] { return launch_rec_backward(up, s); }));
      const AdamHyper hyp = adamHyper(h, parity);
      HIPCK(timed(h, "gemm16_dx_segment", s, [&] { return launch_gemm(GEMM_ROLE_DX, h->dProbs + sb.segDxIdx, 1, sb.segDxBlocks, h->sc, hyp, nullptr, s); }));
      HIPCK(timed(h, "rec_backward_lower", s, [&] { return launch_rec_backward(lo, s); }));
      return launchBackward(h, parity, fuseAdam, s);
    }
    const RecArgs ra = recArgs(h, parity);
    if (h->recFused && h->nConv == 0) {      // forward, head and backward of a sample as one workgroup's work
      const HeadArgs ha = headArgs(h, parity);
      if (rec_step_fused_ok(ra, ha)) {
        HIPCK(timed(h, "rec_step_fused", s, [&] { return launch_rec_step_fused(ra, ha, nullptr, s); }));
        return launchBackward(h, parity, fuseAdam, s);
      }
    }
    HIPCK(timed(h, "rec_forward", s, [&] { return launch_rec_forward(ra, s); }));
    int rc = launchHead(h, parity, s); if (rc) return rc;
    HIPCK(timed(h, "rec_backward", s, [&] { return launch_rec_backward(ra, s); }));
    return launchBackward(h, parity, fuseAdam, s);       // no dX problems for this layout: the dW launch only
  }
  if (h->fusedWideOk) {      // the wide fused kernel (fusedw.hip), then the weight gradients
    int rc = launchFusedWide(h, parity, s); if (rc) return rc;
    return launchWeightGrad(h, parity, fuseAdam, s, false, false);
  }
  if (h->stepChainOk) {      // forward chain + head + input-gradient chain (gemm16.hip: step_chain_kernel), then the weight gradients
    int rc = launchStepChain(h, parity, s); if (rc) return rc;
    return launchBackward(h, parity, fuseAdam, s, false, POST_AGG | POST_BETA, false, true);
  }
  int rc = launchForward(h, parity, s); if (rc) return rc;
  rc = launchHead(h, parity, s); if (rc) return rc;
  return launchBackward(h, parity, fuseAdam, s);
}

// A pre-sampled minibatch is discarded: the generator goes back to where it was before that minibatch was drawn.
int dropPresample(hl_learner* h) {
  if (!h->preValid) return HL_OK;
  h->preValid = false;
  if (h->sidePending) { HIPCK(hipStreamWaitEvent(h->stream, h->evSide, 0)); h->sidePending = false; }      // (large batches: drawn on the side stream)
  HIPCK(launch_rng_restore(h->sc, h->stream));
  return HL_OK;
}

// one full step, eager launches on the main stream.  The order is the reference's (RACER::setupTasks, RACER.cpp:81-108):
// train -> processMemoryBuffer (statistics, every 1000th step the whole-buffer passes, removal, counters) -> gradient
// exchange -> Adam -> beta.  With a communicator every step -- eager or replayed -- issues exactly ONE all-reduce
// (gradient || counters), plus one of the 2 dS + 3 moments on every 1000th step, so replicas may mix the two paths freely.
int stepEager(hl_learner* h, const long long* dFlat) {
  hipStream_t s = h->stream;
  const long long k = h->nGradSteps + 1;
  const bool periodic = (k % 1000) == 0;
  const bool exch = exchanging(h);
  // replicas over peer windows: the weight-gradient launch pushes the gradient itself -- unless another collective (the moments of a
  // 1000th step) comes between that launch and the gradient's, taking the sequence number the push would have aimed at
  struct PushScope { hl_learner* h; ~PushScope() { h->pushGrad = false; } } pushScope{h};
  h->pushGrad = h->pushOk && h->xchg.on && !periodic;
  int p = 0, rc;
  if (h->preValid && !dFlat) {     // drawn by the rider of the previous step (large batches: on the side stream)
    p = h->preParity; h->preValid = false;
    if (h->sidePending) { HIPCK(hipStreamWaitEvent(s, h->evSide, 0)); h->sidePending = false; }
  }
  else { rc = dropPresample(h); if (rc) return rc; rc = launchSample(h, 0, dFlat, true, s); if (rc) return rc; }
  const bool evict = evictionDue(h);
  if (h->bigBatch && !dFlat && !exch && !periodic && !evict && ((k + 1) % 1000) != 0 && h->cfg.dataSamplingAlgo == HL_SAMPLE_UNIFORM) {      // (the prioritised samplers' table follows this step's errors)
    // large batches: the sampler of the NEXT step (one workgroup, hundreds of microseconds) runs beside this step's launches; it
    // starts behind everything queued so far (the buffer it fills was the previous step's) and keeps the generator's state for
    // dropPresample.  Nothing of this step changes what it reads (no removal, no whole-buffer pass).
    HIPCK(hipEventRecord(h->evMain, s)); HIPCK(hipStreamWaitEvent(h->sideStream, h->evMain, 0));
    SampleArgs sa = sampleArgs(h, p ^ 1, nullptr, false); sa.backupRng = 1;
    HIPCK(timed(h, "big_sample_ahead", h->sideStream, [&] { return launch_sample(sa, h->sideStream); }));
    HIPCK(hipEventRecord(h->evSide, h->sideStream));
    h->preValid = true; h->preParity = p ^ 1; h->sidePending = true;
  }
  rc = launchMlp(h, p, !exch, s); if (rc) return rc;
  h->lastParity = p;
  if (!periodic && !evict && !exch) return launchPost(h, p, POST_AGG | POST_BETA, s);
  rc = launchPost(h, p, POST_AGG, s); if (rc) return rc;
  if (periodic) {
    rc = launchPeriodicSweep(h); if (rc) return rc;
    rc = launchMoments(h); if (rc) return rc;
    rc = allreduceMoments(h); if (rc) return rc;
    rc = launchMomentsApply(h, false, 10); if (rc) return rc;
  }
  if (evict) { rc = applyRemoval(h); if (rc) return rc; rc = flushPending(h); if (rc) return rc; }
  if (exch) {
    if (wired(h)) { rc = launchPost(h, p, POST_ENCODE, s); if (rc) return rc; }   // counters as of after the removal
    if (h->xchg.on) return xchgAllreduce(h, h->G, (size_t)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS, 0, p);   // + Adam + beta
    rc = allreduceGrad(h); if (rc) return rc;
    rc = launchAdam(h, p); if (rc) return rc;
  }
  return launchPost(h, p, POST_BETA, s);
}

// ---- replayed graph: U steps on one stream, tail work horizontally fused into the MLP kernels.  The graph starts
//      with the minibatch of its first step already in buffer `p0` and leaves the one of the step after its last in
//      buffer (p0 + U) & 1 ----
int captureSteps(hl_learner* h, int U, int p0, GraphSlot* slot, bool notify = false) {
  if (slot->exec) { hipGraphExecDestroy(slot->exec); slot->exec = nullptr; }
  if (slot->graph) { hipGraphDestroy(slot->graph); slot->graph = nullptr; }
  hipStream_t s0 = h->stream;
  HIPCK(hipStreamSynchronize(s0));
  if (h->bigBatch) { HIPCK(hipStreamSynchronize(h->sideStream)); h->sidePending = false; }      // (it joins the capture below; what an eager step drew ahead is complete: nobody waits on evSide, which the capture re-records)
  HIPCK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  int rc = HL_OK;
  struct PushScope { hl_learner* h; ~PushScope() { h->pushGrad = false; } } pushScope{h};
  h->pushGrad = h->pushOk && h->xchg.on;         // (replayed steps never contain another collective than their gradient's)
  const long long nColl0 = h->nCollectives;      // captured calls are counted when the graph is replayed
  for (int j = 0; j < U && !rc; ++j) {
    const int p = (p0 + j) & 1;
    if (h->plainGraph) {
      // nets whose launches take no riders (recurrent layers behind convolutions, an RNN encoder under MGU layers): stepEager's plain
      // step as graph nodes -- the sampler of the NEXT step first (nothing of this step changes what it reads; it keeps the generator's
      // state for dropPresample), then the step's launch list and its bookkeeping.  12 - 20 launches per step lose their host latency.
      SampleArgs sa = sampleArgs(h, p ^ 1, nullptr, false); sa.backupRng = 1;
      if (launch_sample(sa, s0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "sampler of the next step"); break; }
      rc = launchMlp(h, p, true, s0); if (rc) break;
      rc = launchPost(h, p, POST_AGG | POST_BETA, s0); if (rc) break;
      continue;
    }
    if (h->bigBatch) {
      // large batches (stepEager's plain step as graph nodes): the sampler of the NEXT step -- one workgroup, hundreds of microseconds --
      // is a branch of the graph beside this step's launches (the side stream joins the capture through the event), joined in front
      // of the next step; it keeps the generator's state for dropPresample.  Worth 2 - 4 % at 2048 - 4096 (131.5 -> 126.5, 154.0 -> 151.3 us per
      // step), nothing at 16384: the steps are the sum of their kernels, not launch-bound.  The bookkeeping (25 - 32 us in two launches,
      // needing the head's write-backs only) as a second branch beside the backward launches returned nothing (156.2 against 156.0 us
      // at 4096: a graph branch costs what it overlaps) and was removed.
      if (hipEventRecord(h->evMain, s0) != hipSuccess || hipStreamWaitEvent(h->sideStream, h->evMain, 0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "fork of the sampler branch"); break; }
      SampleArgs sa = sampleArgs(h, p ^ 1, nullptr, false); sa.backupRng = 1;
      if (launch_sample(sa, h->sideStream) != hipSuccess || hipEventRecord(h->evSide, h->sideStream) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "sampler branch"); break; }
      rc = launchMlp(h, p, true, s0); if (rc) break;
      rc = launchPost(h, p, POST_AGG | POST_BETA, s0); if (rc) break;
      if (hipStreamWaitEvent(s0, h->evSide, 0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "join of the sampler branch"); break; }
      continue;
    }
    const bool twoKernel = h->fusedOk || h->fusedWideOk || h->stepChainOk;
    if (twoKernel) {
      // one replica: the far-policy count and the beta update of every step but the last are taken out of the dW launch's
      // bookkeeping rider, where they sat at the end of the kernel's longest workgroup, into a rider of the next fused kernel
      const bool single = !exchanging(h) && !h->noDeferBeta;
      // (the chained step with a few state components: the whole sampler of the next minibatch as its rider -- a 10 us chain beside a 30 us
      //  launch --, the bookkeeping alone on the dW launch of the generic steps; wider states keep the gather helpers of launchWeightGrad)
      const bool chainNarrow = h->stepChainOk && !exchanging(h) && !h->preproc && 2ll * h->B * h->dS <= 16384;
      rc = h->fusedOk ? launchFused(h, p, s0, true, single && j > 0) : (h->fusedWideOk ? launchFusedWide(h, p, s0, true, single && j > 0) : launchStepChain(h, p, s0, true, chainNarrow, single && j > 0)); if (rc) break;
      if (chainNarrow) {
        rc = launchBackward(h, p, true, s0, true, POST_AGG | POST_BETA | (single && j + 1 < U ? POST_DEFER : 0), false, true); if (rc) break;
        continue;
      }
      if (!exchanging(h)) {
        rc = launchWeightGrad(h, p, true, s0, true, true, POST_AGG | POST_BETA | (single && j + 1 < U ? POST_DEFER : 0)); if (rc) break;
        continue;
      }
      // replicas: the exchange is part of the replayed graph (RCCL calls are captured like kernels).  ONE collective per
      // step: the bookkeeping rider of the dW launch appends the four counters to the gradient buffer (four exact 16-bit
      // chunks each), the pass after Adam decodes their sums.
      if (foldUsable(h, p)) {      // the exchange, Adam and the closing bookkeeping inside the weight-gradient launch: two launches per step
        rc = launchWeightGrad(h, p, false, s0, true, true, POST_AGG, true); if (rc) break;
        continue;
      }
      rc = launchWeightGrad(h, p, false, s0, true, true, POST_AGG);
      if (!rc && h->xchg.on) { rc = xchgAllreduce(h, h->G, (size_t)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS, 0, p); if (rc) break; continue; }
      if (!rc) rc = allreduceGrad(h);
      if (!rc) rc = launchAdam(h, p);
      if (!rc) rc = launchPost(h, p, POST_BETA, s0);
      if (rc) break;
      continue;
    }
    // networks off the fused path; replicas: the weight-gradient launches leave the gradient (bookkeeping rider: aggregates and
    // the counters message only), then the exchange with Adam and the rest of the bookkeeping -- as in the fused branch
    const bool exch = exchanging(h);
    if (h->recurrent) {      // window forward, head (+ the sampler of the next step), BPTT, weight gradients (+ bookkeeping)
      const RecArgs ra = recArgs(h, p);
      const HeadArgs ha = headArgs(h, p);
      if (h->recFused && rec_step_fused_ok(ra, ha)) {
        // one launch per sample chain; the draws and sort of the next minibatch ride it (they rode the head launch)
        const ExtraArgs exS = extraSample(h, p ^ 1, PH_A | PH_B);
        if (launch_rec_step_fused(ra, ha, &exS, s0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "rec_step_fused"); break; }
      } else {
        if (launch_rec_forward(ra, s0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "rec_forward"); break; }
        rc = launchHead(h, p, s0, true); if (rc) break;
        if (launch_rec_backward(ra, s0) != hipSuccess) { rc = fail(h, HL_ERR_HIP, "rec_backward"); break; }
      }
    } else {
      rc = launchForward(h, p, s0, true); if (rc) break;
      rc = launchHead(h, p, s0, true); if (rc) break;
    }
    rc = launchBackward(h, p, !exch, s0, true, exch ? POST_AGG : (POST_AGG | POST_BETA), h->recurrent); if (rc) break;
    if (exch) {
      if (h->xchg.on) { rc = xchgAllreduce(h, h->G, (size_t)h->nParams + CNT_MSG_OFFSET + CNT_MSG_FLOATS, 0, p); if (rc) break; continue; }
      rc = allreduceGrad(h);
      if (!rc) rc = launchAdam(h, p);
      if (!rc) rc = launchPost(h, p, POST_BETA, s0);
      if (rc) break;
    }
  }
  if (!rc && notify && launch_notify(h->sc, h->notifyPin, s0) != hipSuccess) rc = fail(h, HL_ERR_HIP, "notify");
  hipError_t e = hipStreamEndCapture(s0, &slot->graph);
  h->nCollectives = nColl0;
  if (rc) return rc;
  if (e != hipSuccess) return hipFail(h, e, "hipStreamEndCapture");
  HIPCK(hipGraphInstantiate(&slot->exec, slot->graph, nullptr, nullptr, 0));
  // the first launch of an executable graph otherwise pays for its upload (~40 us seen on a 20-step call right after the
  // capture); a failure here only means the first launch is slower
  if (hipGraphUpload(slot->exec, s0) != hipSuccess) (void)hipGetLastError();
  slot->steps = U;
  return HL_OK;
}

void invalidateGraphs(hl_learner* h) {
  auto drop = [](GraphSlot& g) {
    if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (g.graph) { hipGraphDestroy(g.graph); g.graph = nullptr; }
  };
  for (auto& gp : h->graphs) for (auto& g : gp) drop(g);
  for (auto& kv : h->exactGraphs) for (auto& g : kv.second) drop(g);      // (the sizes stay: captureAllGraphs re-captures them)
}

// sizes usable by this learner: with a communicator attached the graphs stay short (<= 64 steps, i.e. 64 captured
// collectives: a 2 ms replay already amortises the launch, and nothing here depends on how many collective nodes the
// installed RCCL is comfortable with in one graph); the 999-step graph only ever starts right after a 1000th-step
// sweep, i.e. with buffer 0
// the address translations of the whole replay and of the parameter arrays resident before a stepping phase (touch_kernel)
int touchReplay(hl_learner* h) {
  TouchArgs ta{}; const long long cap = h->capSlots, nE = (long long)h->order.size();
  auto add = [&](const void* p, long long bytes, int stride) { if (p && bytes > 0 && ta.n < 24) { ta.ptr[ta.n] = p; ta.bytes[ta.n] = bytes; ta.stride[ta.n] = stride; ++ta.n; } };
  add(h->rp.S, cap * h->dS * 4, 4096); add(h->rp.A, cap * h->dA * 8, 4096); add(h->rp.MU, cap * h->polDim * 8, 4096); add(h->rp.R, cap * 8, 4096);
  add(h->rp.V, cap * 4, 4096); add(h->rp.ADV, cap * 4, 4096); add(h->rp.RET, cap * 4, 4096); add(h->rp.DQ, cap * 4, 4096); add(h->rp.IMPW, cap * 4, 4096); add(h->rp.DKL, cap * 4, 4096);
  // the per-episode tables the sampler searches and the bookkeeping pass gathers from: every line (a few hundred KB: a fresh
  // learner otherwise warms them by its own random probes over its first dozens of steps -- tools/first_call4.py)
  add(h->rp.posRec, (nE + 1) * (long long)sizeof(PosRec), 64); add(h->rp.posPrefix, (nE + 1) * 8, 64); add(h->rp.posEid, nE * 4, 64);
  add(h->rp.epAgg, (long long)h->nextEid * AGG_N * 4, 64); add(h->rp.epOff, (long long)h->nextEid * 8, 64); add(h->rp.epN, (long long)h->nextEid * 4, 64);
  add(h->rp.epTerm, (long long)h->nextEid, 64); add(h->rp.epTag, (long long)h->nextEid * 8, 64);
  add(h->W, h->nParams * 4, 64); add(h->M1, h->nParams * 4, 64); add(h->M2, h->nParams * 4, 64); add(h->G, h->nParams * 4, 64);
  ta.sink = h->G + h->nParams + 200;
  HIPCK(launch_touch(ta, h->stream));
  return HL_OK;
}

// graph of exactly n steps for both starting buffers, last node = the completion stamp (hl_prepare_steps)
int prepareExact(hl_learner* h, int n) {
  if (!h->useGraph || n >= 1000 || (exchanging(h) && (!(h->exchGraph && wired(h)) || n > 64)) || h->cfg.dataSamplingAlgo != HL_SAMPLE_UNIFORM) return HL_OK;
  if (h->bigBatch && (n > 64 || exchanging(h))) return HL_OK;
  if (!graphUsable(h, n, 0)) return HL_OK;      // (the node cap of the time-step-major recurrent nets holds for exact graphs too: ADVICE r05)
  { const int rc = ensureConvPrep(h); if (rc) return rc; }      // outside the capture: the captured forward refuses stale filter layouts
  if (!h->notifyPin) { HIPCK(hipHostMalloc((void**)&h->notifyPin, 64, hipHostMallocDefault)); *h->notifyPin = 0; }
  if (h->graphsStale) { invalidateGraphs(h); h->graphsStale = false; }
  if (h->exactGraphs.size() >= 8 && !h->exactGraphs.count(n)) {      // a handful of call sizes at most
    auto victim = h->exactGraphs.begin();
    for (auto& g : victim->second) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
    h->exactGraphs.erase(victim);
  }
  auto& slots = h->exactGraphs[n];
  for (int p0 = 0; p0 < 2; ++p0) {
    if (slots[p0].exec) continue;
    const int rc = captureSteps(h, n, p0, &slots[p0], true);
    if (rc) { h->exactGraphs.erase(n); if (!exchanging(h)) return rc; h->err.clear(); (void)hipGetLastError(); return HL_OK; }
  }
  return HL_OK;
}

bool graphUsable(const hl_learner* h, int U, int p0) {
  if (exchanging(h) && U > 64) return false;
  if (h->bigBatch && (U > 64 || exchanging(h))) return false;      // (a dozen nodes and a branch per step; replicas with large local batches step eagerly)
  if (U == 999 && p0 != 0) return false;
  // time-step-major recurrent layers: 2 (LSTM) or 4 (MGU) launches per layer and window step.  The runtime instantiated 83 k nodes in a
  // chain (999 steps of 2 LSTM layers, 18 window steps) and died on 156 k (the same with MGU layers): 64 k nodes at most
  if (h->recTm && (long long)U * (h->recK * h->cfg.n_hidden * (h->cfg.nn_type == HL_NN_MGU ? 4 : 2) + 12) > 65536) return false;
  return true;
}

// Every size and starting buffer is captured at once -- by hl_initialize, or by the first replay after something
// invalidated the graphs (they survive appends and evictions, only a reallocation of the replay invalidates them) -- so
// that no later call pays for a capture in the middle of a training phase.  If RCCL cannot be captured on this system,
// replicas fall back to eager launches for good (the graph is only an optimisation).
int captureAllGraphs(hl_learner* h) {
  constexpr int NS = (int)(sizeof(GRAPH_SIZES) / sizeof(GRAPH_SIZES[0]));
  static_assert(NS <= 16, "hl_learner::graphs is too small");
  if (!h->useGraph || (exchanging(h) && !(h->exchGraph && wired(h)))) return HL_OK;
  { const int rc = ensureConvPrep(h); if (rc) return rc; }
  for (int j = 0; j < NS; ++j) for (int p0 = 0; p0 < 2; ++p0) {
    if (!graphUsable(h, GRAPH_SIZES[j], p0) || h->graphs[j][p0].exec) continue;
    const int rc = captureSteps(h, GRAPH_SIZES[j], p0, &h->graphs[j][p0]);
    if (rc == HL_OK) continue;
    if (!exchanging(h)) return rc;
    // (said once, aloud: from here on this replica pays the host's launch latency for every kernel of every step)
    fprintf(stderr, "smarties_hip: replica %d: the steps' graph could not be captured (%s): stepping with eager launches from now on\n", h->cfg.rank, h->err.c_str());
    h->exchGraph = false; h->err.clear(); (void)hipGetLastError(); invalidateGraphs(h);
    return HL_OK;
  }
  std::vector<int> sizes; for (auto& kv : h->exactGraphs) if (!kv.second[0].exec) sizes.push_back(kv.first);
  for (int n : sizes) { const int rc = prepareExact(h, n); if (rc) return rc; }
  return HL_OK;
}

// run as many plain steps as possible (<= avail) from one graph replay; returns steps done (0 = none)
int replaySteps(hl_learner* h, long long avail, int* done, bool wholeCall = false) {
  *done = 0;
  constexpr int NS = (int)(sizeof(GRAPH_SIZES) / sizeof(GRAPH_SIZES[0]));
  if (!h->graphs[NS - 1][0].exec) { int rc = captureAllGraphs(h); if (rc) return rc; if (!h->graphs[NS - 1][0].exec) return HL_OK; }
  const int p0 = h->preValid ? h->preParity : 0;
  if (h->sidePending) { HIPCK(hipStreamWaitEvent(h->stream, h->evSide, 0)); h->sidePending = false; }      // (large batches: a minibatch drawn beside an eager step)
  if (wholeCall) {      // the whole call as one launch, completion stamp behind it
    auto it = h->exactGraphs.find((int)avail);
    if (it != h->exactGraphs.end() && it->second[p0].exec) {
      const int U = (int)avail;
      if (!h->preValid) { int rc = launchSample(h, 0, nullptr, true, h->stream); if (rc) return rc; }
      HIPCK(hipGraphLaunch(it->second[p0].exec, h->stream));
      h->notifyIssued += 1; h->tailNotify = true;
      h->lastParity = (p0 + U - 1) & 1; h->preValid = true; h->preParity = (p0 + U) & 1;
      if (wired(h)) h->nCollectives += U;
      *done = U;
      return HL_OK;
    }
  }
  if (h->eagerChain > 0 && avail <= h->eagerChain && (h->fusedOk || h->fusedWideOk) && !exchanging(h)) {
    // short calls: the same two launches per step (riders included) issued directly -- no graph launch latency, no
    // first-launch cost of a graph that has not run yet
    if (!h->preValid) { int rc = launchSample(h, 0, nullptr, true, h->stream); if (rc) return rc; }
    const int U = (int)avail;
    for (int j = 0; j < U; ++j) {
      const int p = (p0 + j) & 1;
      int rc = h->fusedOk ? launchFused(h, p, h->stream, true) : launchFusedWide(h, p, h->stream, true); if (rc) return rc;
      rc = launchWeightGrad(h, p, true, h->stream, true, true); if (rc) return rc;
    }
    h->lastParity = (p0 + U - 1) & 1; h->preValid = true; h->preParity = (p0 + U) & 1;
    *done = U;
    return HL_OK;
  }
  for (int i = 0; i < NS; ++i) {
    const int U = GRAPH_SIZES[i];
    if (avail < U || !graphUsable(h, U, p0)) continue;
    if (!h->preValid) { int rc = launchSample(h, 0, nullptr, true, h->stream); if (rc) return rc; }   // first minibatch: its own launch
    HIPCK(hipGraphLaunch(h->graphs[i][p0].exec, h->stream));
    h->lastParity = (p0 + U - 1) & 1;
    h->preValid = true; h->preParity = (p0 + U) & 1;
    if (wired(h)) h->nCollectives += U;
    *done = U;
    return HL_OK;
  }
  return HL_OK;
}

}  // namespace
