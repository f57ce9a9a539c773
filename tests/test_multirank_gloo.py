"""N > 1 replica protocol on CPU: world_size-2 `gloo` processes drive the split-step entry points
through smarties_amd.dist_host, each replica being a CPU oracle learner (ol_* has the signatures of
hl_*).  Checks the order of operations of SURVEY.md 8(e): batch and replay split, gradient sum,
counter sum, moment sum, identical weights on every replica -- against a single-process emulation
that sums the same buffers by hand.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

N_EP, N_STEPS = 40, 1003          # crosses the 1000th step: moments exchange + whole-buffer sweep
CFG = dict(dimS=5, dimA=2, bounded=(1, 0), hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)


def _make(rank, world):
    from oracle_api import oracle_learner, synth_cfg, synth_episode
    from smarties_amd import capi
    L = oracle_learner(capi.make_config(n_ranks=world, rank=rank, **CFG))
    L.init_weights()
    sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)
    for e in range(rank, N_EP, world):          # round-robin split of the episodes over the replicas
        L.append_episode(**synth_episode(sc, e))
    return L


def _worker(rank, world, port, outdir, stale=False):
    import torch.distributed as dist
    from smarties_amd import dist_host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _make(rank, world)
    dist_host.init_replica_weights(L, dist)
    sums = list(dist_host.initialize_host_exchange(L, dist))
    dist_host.step_host_exchange(L, dist, N_STEPS - 400, stale=sums if stale else None)
    dist_host.step_host_exchange(L, dist, 400, stale=sums if stale else None)      # (the one-behind sums carry over between calls)
    w, m1, m2 = L.get_params()
    sc = L.scalars()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), w=w, m1=m1, m2=m2, beta=sc.beta, nGrad=sc.nGradSteps,
             scaling=np.concatenate([np.ravel(x) for x in L.get_scaling()]))
    dist.barrier()
    dist.destroy_process_group()


def _emulate(world, stale=False):
    """Same protocol, one process: the collectives are numpy sums over the replicas.  stale: every replica stores the sums of the
    reduction BEFORE the current one (the reference's other timing, dist_host.step_host_exchange)."""
    Ls = [_make(r, world) for r in range(world)]
    w0 = Ls[0].get_params()[0]
    for L in Ls:
        w, m1, m2 = L.get_params()
        L.set_params(w0, m1, m2)
        L.initialize_begin()
    c = np.sum([L.counters_fetch() for L in Ls], axis=0)
    m = np.sum([L.moments_fetch() for L in Ls], axis=0)
    for L in Ls:
        L.counters_store(c); L.moments_store(m); L.initialize_end()
    c_prev, m_prev = c, m
    for _ in range(N_STEPS):
        for L in Ls:
            L.step_begin()
        g = np.sum([L.grad_fetch() for L in Ls], axis=0, dtype=np.float32)
        ms = [L.moments_fetch() for L in Ls]
        c = np.sum([L.counters_fetch() for L in Ls], axis=0)
        for L, m in zip(Ls, ms):
            L.grad_store(g)
            if m is not None:
                L.moments_store(m_prev if stale else np.sum(ms, axis=0))
            L.counters_store(c_prev if stale else c)
            L.step_end()
        c_prev = c
        if ms[0] is not None:
            m_prev = np.sum(ms, axis=0)
    return Ls


def test_two_replicas_over_gloo_match_single_process_emulation():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(2, port, td), nprocs=2, join=True)
        r = [np.load(os.path.join(td, "r%d.npz" % i)) for i in range(2)]
    # replicas stay bit-identical (same summed gradient, same Adam)
    for k in ("w", "m1", "m2", "beta", "nGrad"):
        assert np.array_equal(r[0][k], r[1][k]), k
    assert int(r[0]["nGrad"]) == N_STEPS
    Ls = _emulate(2)
    w, m1, m2 = Ls[0].get_params()
    # gloo's 2-rank sum is a+b in either order: bit-identical to the numpy sum
    assert np.array_equal(w, r[0]["w"]) and np.array_equal(m1, r[0]["m1"]) and np.array_equal(m2, r[0]["m2"])
    assert Ls[0].scalars().beta == float(r[0]["beta"])
    for i, L in enumerate(Ls):
        sc = np.concatenate([np.ravel(x) for x in L.get_scaling()])
        assert np.array_equal(sc, r[i]["scaling"])


def test_two_replicas_over_gloo_one_reduction_behind():
    """dist_host.step_host_exchange(stale=...): the reference's other reduction timing over gloo (two processes, two calls: the sums
    carry over) equals the single-process emulation of it -- and is not the current-sums trajectory."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(2, port, td, True), nprocs=2, join=True)
        r = [np.load(os.path.join(td, "r%d.npz" % i)) for i in range(2)]
    for k in ("w", "m1", "m2", "beta", "nGrad"):
        assert np.array_equal(r[0][k], r[1][k]), k
    Ls = _emulate(2, stale=True)
    assert np.array_equal(Ls[0].get_params()[0], r[0]["w"]) and Ls[0].scalars().beta == float(r[0]["beta"])
    for i, L in enumerate(Ls):
        assert np.array_equal(np.concatenate([np.ravel(x) for x in L.get_scaling()]), r[i]["scaling"])
    assert _emulate(2)[0].scalars().beta != float(r[0]["beta"]) or not np.array_equal(_emulate(2)[0].get_params()[0], r[0]["w"])


def test_split_sizes_follow_reference_rule():
    """Settings/HyperParameters.cpp:186-197: batch and replay budget divided by the replica count."""
    L1, L2 = _make(0, 1), _make(0, 2)
    assert L1.B == CFG["batchSize"] and L2.B == CFG["batchSize"] // 2


def test_moments_exchange_only_on_sweep_steps():
    L = _make(0, 2)
    L.initialize()
    L.step_begin()
    assert L.moments_fetch() is None      # not a 1000th step: HL_ERR_STATE, nothing pending
    g = L.grad_fetch(); L.grad_store(g)
    L.counters_store(L.counters_fetch())
    L.step_end()
    assert L.scalars().nGradSteps == 1
