// smarties_amd/csrc/learner_xchg.h -- part of learner.cpp's ONE translation unit (included there, like step_exec.h): replicas: the RCCL communicator (hl_comm_*) and the peer-window exchange's set-up (hl_xchg_export / hl_xchg_connect; kernels: xchg.hip)
#pragma once

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
    // (one replica per device: 32 chunks -- measured in loopback with caps of 16 / 24 / 32 / 48 / 64: 34.3 / 32.7 / 32.7 / 33.3 / 35.0 us per step at
    //  2 replicas, 37.6 / 33.9 / 33.2 / 33.9 / 34.9 at 8: more workgroups lengthen the two-phase wait, fewer the sums and Adam slices)
    x.maxChunks = (most <= 1 || (h->generic & 1024)) ? XCHG_CHUNKS_NODE : std::max(4, XCHG_CHUNKS / most); }      // (GENERIC & 1024: a node's cut on a shared device -- tools/replica_loopback.py, where only ONE replica steps)
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
