"""Round 6, on the GPU: the replicas' gradient exchange (xchg.hip, the dW launches' peer-window push, the exchange folded into the weight-
gradient launch) at the shapes BASELINE.json quotes -- VERDICT r05: until now every exchange test ran 2 x 32-unit nets with a 6 KB message.
Reference: the gradient's MPI_Iallreduce (Network/Optimizer.cpp:110-132), the counters' and moments' reductions
(Utils/DelayedReductor.cpp:53-83), batch and replay budget split over the learners (Settings/HyperParameters.cpp:186-197).

All replicas live on this one GPU (a thread each, windows as plain pointers): the wire protocol, chunking, arrival stamps, rank-order
sums and Adam slices are those of a node; the links are not."""
import ctypes as C

import numpy as np
import pytest

from smarties_amd import capi
from oracle_api import synth_cfg, synth_episode
from test_hip_parity import hip_learner, _both


def _replicas(hip_api, cfg_kw, sc, n_ranks, n_eps, connect):
    """n_ranks replicas holding disjoint episodes (e = r mod n_ranks), common start weights, statistics of the local shards (the start
    of the host-exchange run), then connected through each other's windows or left to the host-exchange entry points."""
    Ls = []
    for r in range(n_ranks):
        L = hip_learner(hip_api, capi.make_config(n_ranks=n_ranks, rank=r, **cfg_kw))
        L.init_weights()
        for e in range(r, n_eps, n_ranks):
            L.append_episode(**synth_episode(sc, e, cfg_kw.get("n_options", 0)))
        Ls.append(L)
    w0 = Ls[0].get_params()[0]
    for L in Ls:
        w, m1, m2 = L.get_params(); L.set_params(w0, m1, m2); L.initialize()
    if connect:
        handles = [L.xchg_export() for L in Ls]
        _both(Ls, lambda L: L.xchg_connect(handles))
    return Ls


def _host_step(Ls):
    """One step of every replica with the sums formed on the host in RANK ORDER in fp32 -- ((g0 + g1) + g2) + ... -- which is what every
    replica's exchange kernel computes (all replicas then hold the same bits)."""
    for L in Ls:
        L.step_begin()
    gs = [L.grad_fetch() for L in Ls]
    g = gs[0].copy()
    for q in gs[1:]:
        g = (g + q).astype(np.float32)
    ms = [L.moments_fetch() for L in Ls]
    c = np.sum([L.counters_fetch() for L in Ls], axis=0)
    m = None
    if ms[0] is not None:
        m = ms[0].copy()
        for q in ms[1:]:
            m = m + q
    for L in Ls:
        L.grad_store(g)
        if m is not None:
            L.moments_store(m)
        L.counters_store(c)
        L.step_end()


def _assert_same(X, H, where):
    for r in range(len(X)):
        for name, a, b in zip(("W", "M1", "M2"), X[r].get_params(), H[r].get_params()):
            if not np.array_equal(a, b):
                d = np.nonzero(a != b)[0]
                raise AssertionError("after %d steps, replica %d, %s: %d of %d elements differ (first %s, last %s), largest difference %.3e" % (
                    where, r, name, d.size, a.size, d[:6].tolist(), d[-3:].tolist(), float(np.abs(a - b).max())))
        sx, sh = X[r].scalars(), H[r].scalars()
        assert sx.beta == sh.beta and sx.nFarPolicySteps == sh.nFarPolicySteps, (where, r)
        assert np.array_equal(X[r].get_rng_state(), H[r].get_rng_state()), (where, r)
    for r in range(1, len(X)):
        assert np.array_equal(X[0].get_params()[0], X[r].get_params()[0]), (where, r)      # replicas identical


def _exchange_parity(hip_api, cfg_kw, sc, n_ranks, n_eps, calls):
    X = _replicas(hip_api, cfg_kw, sc, n_ranks, n_eps, True)
    H = _replicas(hip_api, cfg_kw, sc, n_ranks, n_eps, False)
    assert X[0].B == cfg_kw["batchSize"] // n_ranks
    done = 0
    for n in calls:
        _both(X, lambda L: (L.step(n), L.sync()))
        for _ in range(n):
            _host_step(H)
        done += n
        _assert_same(X, H, done)
    coll = hip_api.lib.hl_debug_collectives
    coll.restype = C.c_int64; coll.argtypes = [C.c_void_p]
    n0 = coll(X[0].h)
    assert all(coll(L.h) == n0 for L in X) and n0 >= done      # one collective per step (+ the moments' on a 1000th step)
    for L in X + H:
        L.close()
    return n0


CALLS_1005 = (1, 1, 3, 20, 70, 900, 10)      # eager calls, replayed graphs, the 1000th-step sweep with its moments exchange


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 8])
@pytest.mark.parametrize("route", ["pushed", "unpushed"])
def test_replica_exchange_at_the_north_star_shape(hip_api, monkeypatch, n_ranks, route):
    """cfg-NS (17 states, 6 bounded actions, 2 x 256 SoftSign, GLOBAL batch 256: 72 976 parameters = a 292 KB message in 64 chunks, 354 dW
    tiles pushing from their epilogues) over 2 and 8 replicas: weights, both Adam moments, beta, the far-policy count and the generator
    bit-equal to the host-formed rank-order sums after eager calls, replayed graphs and the 1000th step; replicas identical.  Routes:
    the gradient pushed by the dW tiles' epilogues with the exchange launch behind it (default), and the exchange kernel's own push
    (SMARTIES_HIP_NO_PUSH=1).  The folded two-launch step: test_folded_replica_step_in_a_fresh_process."""
    if route == "unpushed":
        monkeypatch.setenv("SMARTIES_HIP_NO_PUSH", "1")
    cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
    _exchange_parity(hip_api, cfg_kw, sc, n_ranks, 40 * n_ranks, CALLS_1005)


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,batch", [(8, 128), (4, 128)], ids=["8xB16", "4xB32"])
@pytest.mark.parametrize("route", ["pushed", "unpushed"])
def test_replica_exchange_at_the_humanoid_shape(hip_api, monkeypatch, route, n_ranks, batch):
    """BASELINE config 3 (Humanoid-v2 through apps/OpenAI_gym/HumanoidWrapper.py under 8 learner replicas: 257 observed states, 17 actions,
    2 x 256, 32 samples per replica) on fused_wide_kernel: the same bit-equality over 1005 steps.  The configuration itself -- 8 replicas x 32 samples -- cannot
    run on ONE device: at 257 states a workgroup of the fused kernel owns its CU's LDS, 8 x (2 panels x 16 + 8 rider slots) = 320
    workgroups want 256 CUs, and the panel groups of different replicas then wait inside their kernels for CUs held by each other's
    (bounded spins, device error 77: seen).  On a node every replica has 256 CUs of its own.  Here: 8 replicas x 16 samples (192
    workgroups) and 4 replicas x 32 samples (160), which cover the replica count and the per-replica shape separately."""
    if route == "unpushed":
        monkeypatch.setenv("SMARTIES_HIP_NO_PUSH", "1")
    cfg_kw = dict(dimS=257, dimA=17, hidden=(256, 256), nnFunc="SoftSign", batchSize=batch, maxTotObsNum=65536, randSeed=9)
    sc = synth_cfg(seed=13, dimS=257, dimA=17, lenMin=30, lenMax=120, pTerm=0.3)
    _exchange_parity(hip_api, cfg_kw, sc, n_ranks, 30 * n_ranks, CALLS_1005)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,n_eps,calls", [
    (dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=4096, maxTotObsNum=262144, randSeed=42), 600, (1, 2, 5, 990, 4)),
    (dict(dimS=6, dimA=2, bounded=[1, 0], hidden=(32, 32), nnFunc="Tanh", batchSize=2560, maxTotObsNum=200000, randSeed=37,
          nn_type=capi.NN_LSTM, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=4), 800, (1, 2, 6)),
    (dict(dimS=576, dimA=2, nAppendedObs=0, conv=[(12, 12, 4, 8, 3, 1), (10, 10, 8, 16, 4, 2)], hidden=(32,), nnFunc="Tanh",
          batchSize=2400, maxTotObsNum=60000, randSeed=43), 500, (1, 2, 4)),
    (dict(dimS=6, dimA=2, bounded=[1, 0], hidden=(32, 32), nnFunc="Tanh", batchSize=64, maxTotObsNum=20000, randSeed=37,
          nn_type=capi.NN_LSTM, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=4), 120, (1, 2, 30, 70)),
    (dict(dimS=576, dimA=2, nAppendedObs=0, conv=[(12, 12, 4, 8, 3, 1), (10, 10, 8, 16, 4, 2)], hidden=(32,), nnFunc="Tanh",
          batchSize=64, maxTotObsNum=20000, randSeed=43), 120, (1, 2, 30, 70))],
    ids=["dense-2x256-local2048", "lstm-2x32-local1280", "conv-local1200", "lstm-2x32-local32", "conv-local32"])
def test_two_replicas_with_large_local_batches_and_other_layer_types(hip_api, cfg_kw, n_eps, calls):
    """The combination ADVICE r04 found broken and learner.cpp now gates (`pushOk = ... && !bigBatch`): two replicas whose LOCAL batch is
    above 1024 -- the 64 x 64 weight-gradient tiles over row chunks and the split-row joins never push, the exchange kernel sends their
    gradient -- for the bench network (across a 1000th step), an LSTM net and a convolutional net; the same two nets at small local
    batches (per-sample recurrent kernels, filter-gradient partials + conv_reduce_adam: their gradient also leaves from the exchange kernel)."""
    lenMax = 200 if cfg_kw["dimS"] == 17 else 40
    sc = synth_cfg(seed=41, dimS=cfg_kw["dimS"], dimA=cfg_kw["dimA"], lenMin=4, lenMax=lenMax, pTerm=0.4)
    _exchange_parity(hip_api, cfg_kw, sc, 2, n_eps, calls)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw", [dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42),
                                    dict(dimS=257, dimA=17, hidden=(256, 256), nnFunc="SoftSign", batchSize=32, maxTotObsNum=65536, randSeed=9)],
                         ids=["north-star", "humanoid-b32"])
def test_rccl_sequence_at_the_baseline_shapes_on_one_rank(hip_api, cfg_kw):
    """hl_comm_init / ncclAllReduce of `gradient || counters` (292 KB) as far as one device allows: a 1-rank RCCL communicator switches the
    step to the N > 1 sequence (dW without Adam -> all-reduce -> Adam launch -> counters decoded -> beta; the moments' all-reduce on the
    1000th step), eager and captured in the replayed graphs -- bit-equal to the single-replica step.  Network/Optimizer.cpp:110-160."""
    from oracle_api import fill_synth
    sc = synth_cfg(seed=7, dimS=cfg_kw["dimS"], dimA=cfg_kw["dimA"], lenMin=40, lenMax=200, pTerm=0.3)
    A = hip_learner(hip_api, capi.make_config(**cfg_kw)); A.init_weights(); fill_synth(A, sc, 60); A.initialize()
    Bq = hip_learner(hip_api, capi.make_config(**cfg_kw)); Bq.init_weights(); fill_synth(Bq, sc, 60)
    raw = (C.c_uint8 * 128)()
    assert hip_api.fn("comm_unique_id")(raw) == 0
    Bq.comm_init(bytes(raw))
    Bq.initialize()
    for n in (1, 7, 64, 931):
        A.step(n); Bq.step(n)
        for a, b in zip(A.get_params(), Bq.get_params()):
            assert np.array_equal(a, b), n
        assert A.scalars().beta == Bq.scalars().beta and A.scalars().nFarPolicySteps == Bq.scalars().nFarPolicySteps
    assert np.array_equal(A.get_rng_state(), Bq.get_rng_state())


@pytest.mark.gpu
def test_folded_replica_step_in_a_fresh_process():
    """SMARTIES_HIP_FOLD=1: the exchange, Adam and the closing bookkeeping inside the weight-gradient launch -- a replica's replayed step
    is TWO kernels (hl_debug_graph_kernels) where the default step is three -- bit-equal to the host-formed sums over 200 steps of two
    cfg-NS replicas.  In a process of its own: the folded hand-off is only trusted on fresh device memory (learner.cpp, hl_xchg_connect;
    docs/EXPERIMENTS.md), which is why it is not the default."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import os, sys, ctypes as C; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, torch\n"
        "from smarties_amd import capi, load_hip; from oracle_api import synth_cfg; import test_hip_r6 as t6\n"
        "api = load_hip(); gk = api.lib.hl_debug_graph_kernels; gk.restype = C.c_int64; gk.argtypes = [C.c_void_p, C.c_int32]\n"
        "kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc='SoftSign', batchSize=256, maxTotObsNum=65536, randSeed=42)\n"
        "sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)\n"
        "X = t6._replicas(api, kw, sc, 2, 80, True); H = t6._replicas(api, kw, sc, 2, 80, False)\n"
        "print('KERNELS_PER_8_STEPS', gk(X[0].h, 8))\n"
        "for n in (1, 3, 20, 70, 106):\n"
        "    t6._both(X, lambda L: (L.step(n), L.sync()))\n"
        "    for _ in range(n): t6._host_step(H)\n"
        "    t6._assert_same(X, H, n)\n"
        "print('FOLD_OK')\n") % (os.path.dirname(here), here)
    for fold, per8 in (("1", 16), ("0", 24)):
        env = dict(os.environ, SMARTIES_HIP_FOLD=fold, GPU_MAX_HW_QUEUES="16", SMARTIES_HIP_XCHG_TIMEOUT_MS="30000")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert "FOLD_OK" in out.stdout and ("KERNELS_PER_8_STEPS %d" % per8) in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_exchange_windows_survive_generations_of_learners(hip_api):
    """The finding behind learner.cpp's window pool: an exchange window (uncached device memory) that went back to the allocator made
    kernels of LATER learners in the process read stale values -- the host-summed replicas of the next generation computed gradients
    off by whole tiles (7 of 8 generations; once a wild pointer).  Six generations of {two connected replicas + two host-summed ones}:
    every generation's gradients, weights and moments must be those of the first, bit for bit."""
    cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
    first = None
    for gen in range(6):
        X = _replicas(hip_api, cfg_kw, sc, 2, 80, True)
        H = _replicas(hip_api, cfg_kw, sc, 2, 80, False)
        _both(X, lambda L: (L.step(2), L.sync()))
        for L in H:
            L.step_begin()
        gs = [L.grad_fetch() for L in H]
        for L in H:
            L.grad_store((gs[0] + gs[1]).astype(np.float32)); L.counters_store(np.sum([q.counters_fetch() for q in H], axis=0)); L.step_end()
        _host_step(H)
        _assert_same(X, H, 2)
        state = [g.copy() for g in gs] + [a.copy() for a in X[0].get_params()]
        if first is None:
            first = state
        for a, b in zip(state, first):
            assert np.array_equal(a, b), gen
        for L in X + H:
            L.close()
