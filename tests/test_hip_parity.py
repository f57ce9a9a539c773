"""GPU suite (-m gpu): the HIP path, called through the hl_* C-ABI, against
  (1) golden fixtures recorded from the compiled reference (tests/golden/*.bin), and
  (2) the CPU oracle on the same seeded inputs (bit-exact sample indices and ReF-ER masks,
      fp32 quantities to 1e-5 of the infinity norm, fp64 head quantities to 1e-9),
plus size-independent properties at the BASELINE.json replay size (1M transitions)."""
import ctypes as C
import numpy as np
import pytest

from oracle_api import oracle_learner, fill_synth, synth_cfg, synth_episode
from parity import (far_count_loop, storage_order, load_fixture, fixture_config, fixture_synth, fixture_arrival, setup_from_fixture, relinf,
                    episode_arrays_by_tag, fixture_arrays_by_tag, stats_line, lines_agree, fx_vec_dev, flat_for)
from smarties_amd import capi

pytestmark = pytest.mark.gpu

ACT_FIXTURES = ["act_%s.bin" % f for f in ("LRelu", "Sigm", "HardSign", "SoftPlus", "ExpPlus", "Exp")]     # the other names of makeFunction (Functions.h:643-668)
EVICT_FIXTURES = ["evict_%s.bin" % f for f in ("farpolfrac", "maxkldiv", "minerror")]      # ERoldSeqFilter (MemoryProcessing.cpp:261-298)
FUNC_OF = {"hp_odd.bin": "Tanh", "discrete_lstm.bin": "Tanh", "gauss_mgu.bin": "Tanh", "one_layer_relu.bin": "Relu", "deep_tanh.bin": "Tanh", "racer_lstm.bin": "Tanh", "vracer_mgu.bin": "Tanh", **{n: n[4:-4] for n in ACT_FIXTURES}}
TOL32 = 1e-5     # north_star: 1e-5 relative fp32
TOL64 = 1e-9


def hip_learner(hip_api, cfg):
    return capi.Learner(hip_api, cfg)


our_flat_for = flat_for


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "ns_shape.bin", "racer_gauss.bin", "racer_discrete.bin", "racer_lstm.bin", "vracer_mgu.bin", "threads3.bin", "hp_odd.bin", "hp_lowclip.bin", "discrete_lstm.bin", "one_layer_relu.bin", "gauss_mgu.bin", "crowded_sampler.bin", "appended_dense.bin"])
def test_init_weights_and_initialize_match_reference(hip_api, name):
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    setup_from_fixture(L, fx)
    # after initialize: weights are the reference's W0 bit for bit, generator state too
    w, m1, m2 = L.get_params()
    assert np.array_equal(w, fx["W0"]) and not m1.any() and not m2.any()
    assert np.array_equal(L.get_rng_state(), fx["rng0"])
    s = L.scalars()
    assert s.nStoredSteps == int(fx["cfg"][7])
    assert s.beta == fx["beta0"][0] and s.CmaxRet == fx["cmax0"][0]
    m, sc, r = L.get_scaling()
    assert np.allclose(np.concatenate([m, sc, r]), fx["scaling0"], rtol=2e-7, atol=1e-7)
    lens = {e: synth_episode(fixture_synth(fx), e)["rewards"].size for e in range(int(fx["cfg"][3]))}
    mine = episode_arrays_by_tag(L, capi.EP_RETURN)
    ref = fixture_arrays_by_tag(fx, "ret0_tags", "ret0", lens)
    for tag, arr in ref.items():
        assert np.allclose(mine[tag], arr, rtol=2e-6, atol=2e-6), tag


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "ns_shape.bin", "racer_gauss.bin", "racer_discrete.bin", "racer_lstm.bin", "vracer_mgu.bin", "threads3.bin", "hp_odd.bin", "hp_lowclip.bin", "discrete_lstm.bin", "one_layer_relu.bin", "gauss_mgu.bin", "crowded_sampler.bin", "appended_dense.bin"] + ACT_FIXTURES)
def test_steps_follow_reference_fixture(hip_api, name):
    """Feed the (episode, t) pairs the reference sampled at each tapped step and compare every
    per-sample quantity and the summed gradient / Adam update with the reference's own values."""
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    setup_from_fixture(L, fx)
    # the oracle in the reference's storage order (its std::sort over episodes whose IDs all tie reshuffles them every step)
    O = oracle_learner(fixture_config(fx, nnFunc=FUNC_OF.get(name), episode_order=capi.ORDER_REFERENCE))
    setup_from_fixture(O, fx)
    nSteps = int(fx["cfg"][4])
    for k in range(1, nSteps + 1):
        sk = "s%d_" % k
        if sk + "flat" not in fx:
            break
        flat = our_flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        order = np.argsort(flat, kind="stable")          # the library wants sorted indices
        L.step(1, flat=flat[order])
        ref_order = storage_order(O)                     # (the statistics pass of a step runs before its std::sort)
        if name == "appended_dense.bin":                 # (drawn by the harness's RestrictedSampler, oracle/ref_driver.cpp)
            O.step(1, flat=our_flat_for(O, fx[sk + "tag"], fx[sk + "t"]))
        else:
            O.step(1)
        assert np.array_equal(L.readback(capi.TAP_TAG), fx[sk + "tag"][order])
        assert np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"][order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_DELTAQ), fx[sk + "dq"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"][order]) < TOL32
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"][order])
        if sk + "gradSum" in fx:
            assert relinf(L.readback(capi.TAP_GRADSUM), fx[sk + "gradSum"]) < TOL32
        if sk + "W" in fx:
            w, m1, m2 = L.get_params()
            assert relinf(w, fx[sk + "W"]) < TOL32
            assert relinf(m1, fx[sk + "M1"]) < TOL32 and relinf(m2, fx[sk + "M2"]) < 2 * TOL32
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-12 * abs(sca.beta)
        assert sca.CmaxRet == fx["traj_cmax"][k - 1]
        # ReplayStats::nFarPolicySteps is a float-add/truncate loop over the episodes in storage order: the library runs that loop
        # in ITS order (newest first; == the oracle in that order, test_device_sampler_and_update_match_oracle), and over the
        # reference's order its fractions give the reference's number
        assert O.scalars().nFarPolicySteps == fx["traj_nfar"][k - 1]
        assert far_count_loop(L, ref_order) == fx["traj_nfar"][k - 1]
        assert sca.nFarPolicySteps == far_count_loop(L, storage_order(L))


PER_FIXTURES = ["sample_%s.bin" % f for f in ("PERrank", "PERerr", "PERseq")]      # dataSamplingAlgo (Sampling.cpp:101-296)


@pytest.mark.parametrize("algo", ["PERrank", "PERerr", "PERseq"])
def test_prioritised_sampler_tables(hip_api, algo):
    """The discrete distribution behind dataSamplingAlgo PERrank / PERerr / PERseq on ~60 000 stored transitions (the
    one-wavefront chain crosses many 1024-value blocks): probabilities as Sampling.cpp:137-146 / 192-196 / 247-249 define them
    from the errors in the replay, and the cumulative table bit for bit what libstdc++'s discrete_distribution builds --
    sequential double sum, division, sequential partial_sum, last entry 1.
    (The reference's own minibatches, tests/golden/sample_PER*.bin, are reproduced by the oracle in its reference-order mode --
    test_oracle_golden.py; they cannot be fed to the library: the drawn flat indices mean (episode, step) pairs through the
    STORAGE order, where the reference's is an artefact of its non-stable per-step std::sort (as is its ranking of equal errors
    for PERrank) and the library's is newest-first / stable.  Next test: library == oracle in that mode, bit for bit.)"""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=256, maxTotObsNum=100000, randSeed=4, dataSamplingAlgo=algo)
    sc = synth_cfg(seed=13, dimS=5, dimA=2, lenMin=150, lenMax=250, pTerm=0.4)
    G = hip_learner(hip_api, capi.make_config(**cfg_kw))
    G.init_weights(); fill_synth(G, sc, 300); G.initialize()
    f = hip_api.lib.hl_debug_per_table; f.restype = C.c_int64; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    G.step(40)                                                     # errors of ~10 000 transitions are now their own
    nEp = G.scalars().nStoredEps
    dq = [G.episode_field(p, capi.EP_DELTAQ) for p in range(nEp)]      # (N values per episode, the last one unused)
    G.step(1)                                                      # the table this step drew from was built on `dq`
    n = int(G.scalars().nStoredSteps) if algo != "PERseq" else nEp
    prob, cp = np.zeros(n, np.float32), np.zeros(n, np.float64)
    assert f(G.h, prob.ctypes.data, cp.ctypes.data, n) == n
    eps = np.float32(np.finfo(np.float32).eps)
    if algo == "PERseq":
        want = None                                                # (from the running per-episode average: next test)
    else:
        d2 = np.concatenate([d[:-1] * d[:-1] for d in dq]).astype(np.float32)
        if algo == "PERerr":
            want = np.sqrt(np.sqrt(d2 + eps))
        else:
            order = np.argsort(-d2.astype(np.float64), kind="stable")       # decreasing error, equal errors in storage order
            want = np.ones(n, np.float32)
            want[order] = np.where(d2[order] > 0, (1 / np.sqrt(np.sqrt(np.arange(1, n + 1, dtype=np.float64)))).astype(np.float32), np.float32(1))
    if want is not None:
        assert np.array_equal(prob, want.astype(np.float32))
    assert (prob > 0).all()
    total = np.cumsum(prob.astype(np.float64))[-1]                 # (np.cumsum adds in sequence, as std::accumulate does)
    ref = np.cumsum(prob.astype(np.float64) / total); ref[-1] = 1.0
    assert np.array_equal(cp, ref)


@pytest.mark.parametrize("algo,extra", [("PERrank", {}), ("PERerr", {}), ("PERseq", {}),
                                        ("PERerr", dict(nn_type=capi.NN_LSTM, nnFunc="Tanh", nnBPTTseq=6)),                  # recurrent path
                                        ("PERseq", dict(adv_kind=capi.ADV_GAUSSIAN)), ("PERrank", dict(adv_kind=capi.ADV_DISCRETE, n_options=4, dimA=1, bounded=[0]))],   # RACER heads
                         ids=lambda v: v if isinstance(v, str) else "-".join(v) or "vracer")
def test_prioritised_samplers_follow_the_oracle_while_episodes_arrive(hip_api, algo, extra):
    """The same samplers against the oracle on a replay that grows and evicts between steps (the table is rebuilt over the
    current contents each time), batch 64 over ~1500 transitions: duplicates are redrawn as the reference does.  Calls of
    several steps take the eager route (the distribution is rebuilt before every minibatch)."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=64, maxTotObsNum=1500, minTotObsNum=500, randSeed=4,
                  dataSamplingAlgo=algo)
    cfg_kw.update(extra)
    sc = synth_cfg(seed=13, dimS=5, dimA=cfg_kw["dimA"], lenMin=10, lenMax=60, pTerm=0.4)
    G, O = _pair(hip_api, cfg_kw, sc, 40)
    e = 40
    for k in range(25):
        n = 1 if k % 3 else 3
        G.step(n); O.step(n)
        _compare_step(G, O)
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
        for L in (G, O):
            L.append_episode(**synth_episode(sc, e, cfg_kw.get("n_options", 0)))
        e += 1
    assert relinf(G.get_params()[0], O.get_params()[0]) < TOL32


def test_unknown_sampler_is_rejected(hip_api):
    cfg = capi.make_config(dimS=5, dimA=2, hidden=(16,), batchSize=8)
    cfg.dataSamplingAlgo = 7
    with pytest.raises(RuntimeError):
        capi.Learner(hip_api, cfg)


def test_device_sampler_on_the_reference_s_crowded_minibatches(hip_api):
    """Batch 64 out of 93 stored transitions (tests/golden/crowded_sampler.bin): most draws collide, Sample_uniform's sort / unique /
    redraw loop (Sampling.cpp:69-93) runs many rounds per minibatch -- about 105 generator draws for 64 indices.  The device
    sampler, drawing on its own, produces the compiled reference's flat indices and generator states for all 25 steps."""
    fx = load_fixture("crowded_sampler.bin")
    L = hip_learner(hip_api, fixture_config(fx))
    setup_from_fixture(L, fx)
    L.set_tap(True)
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        assert np.array_equal(L.get_rng_state(), fx[sk + "rng"]), k
        L.step(1)
        assert np.array_equal(L.readback(capi.TAP_FLAT), fx[sk + "flat"]), k
    draws = [(int(fx["s%d_rng" % (k + 1)][-1]) - int(fx["s%d_rng" % k][-1])) % 624 for k in range(1, 25)]
    assert min(draws) > 64 + 1                     # (every minibatch needed redraws; + 1: the optimizer's draw)


def test_generator_stream_of_a_reference_run_with_three_threads(hip_api):
    """ref_threads = 3 (tests/golden/threads3.bin, recorded with three OpenMP threads): the reference seeds two more generators from
    the main one, so weights and every minibatch come from a stream shifted by two draws, and each Adam step takes ONE draw from
    it whatever the thread count.  The library sampling on its own walks exactly that stream (the drawn flat indices too; which
    transitions they denote depends on the storage order)."""
    fx = load_fixture("threads3.bin")
    assert int(fx["threads"][0]) == 3
    L = hip_learner(hip_api, fixture_config(fx))
    setup_from_fixture(L, fx)
    assert np.array_equal(L.get_params()[0], fx["W0"]) and np.array_equal(L.get_rng_state(), fx["rng0"])
    L.set_tap(True)
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        if sk + "rng" in fx:
            assert np.array_equal(L.get_rng_state(), fx[sk + "rng"]), k
        L.step(1)
        if sk + "flat" in fx:
            assert np.array_equal(L.readback(capi.TAP_FLAT), fx[sk + "flat"]), k
    one = hip_learner(hip_api, fixture_config(fx, ref_threads=1)); one.init_weights()
    assert not np.array_equal(one.get_params()[0], fx["W0"])               # (one thread: another stream)


def _pair(hip_api, cfg_kw, sc, n_eps):
    G = hip_learner(hip_api, capi.make_config(**cfg_kw))
    O = oracle_learner(capi.make_config(**cfg_kw))
    for L in (G, O):
        L.init_weights()
        fill_synth(L, sc, n_eps)
        L.initialize()
        L.set_tap(True)
    return G, O


def _compare_step(G, O):
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert np.array_equal(G.readback(capi.TAP_TAG), O.readback(capi.TAP_TAG))
    assert np.array_equal(G.readback(capi.TAP_TSTEP), O.readback(capi.TAP_TSTEP))
    assert np.array_equal(G.readback(capi.TAP_STATE), O.readback(capi.TAP_STATE))   # same fp32 op order
    assert relinf(G.readback(capi.TAP_OUTPUT), O.readback(capi.TAP_OUTPUT)) < TOL32
    assert relinf(G.readback(capi.TAP_RHO), O.readback(capi.TAP_RHO)) < TOL32
    assert relinf(G.readback(capi.TAP_DKL), O.readback(capi.TAP_DKL)) < TOL32
    assert relinf(G.readback(capi.TAP_OUTGRAD), O.readback(capi.TAP_OUTGRAD)) < TOL32
    assert np.array_equal(G.readback(capi.TAP_FAR), O.readback(capi.TAP_FAR))
    assert relinf(G.readback(capi.TAP_GRADSUM), O.readback(capi.TAP_GRADSUM)) < TOL32


@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps,steps", [
    (dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42),
     dict(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5), 30, 40),
    (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=100000, randSeed=1),
     dict(seed=9, dimS=17, dimA=6, lenMin=150, lenMax=250, pTerm=0.2), 300, 12),
    (dict(dimS=9, dimA=3, bounded=[0, 0, 0], hidden=(24, 16, 8), nnFunc="Tanh", batchSize=8, maxTotObsNum=1000,
          randSeed=5, nnLambda=1e-4, learnrate=1e-3),
     dict(seed=3, dimS=9, dimA=3, lenMin=3, lenMax=30, pTerm=0.3), 20, 25),
    (dict(dimS=33, dimA=17, bounded=[0] * 17, hidden=(96, 40), nnFunc="Relu", batchSize=48, maxTotObsNum=5000,
          randSeed=8),
     dict(seed=4, dimS=33, dimA=17, lenMin=20, lenMax=80, pTerm=0.1, muSpread=0.2), 60, 10),
    # shapes served by the fused forward/head/dX kernel (fused.hip): one tile per panel (no panel
    # barrier), odd batch, every episode truncated and short -> many next-state rows
    (dict(dimS=3, dimA=1, bounded=[1], hidden=(16, 16), batchSize=5, maxTotObsNum=500, randSeed=3),
     dict(seed=11, dimS=3, dimA=1, lenMin=3, lenMax=6, pTerm=0.0), 25, 30),
    # widest state / action spaces of the fused kernel, Tanh instantiation, partial last panel
    (dict(dimS=32, dimA=7, bounded=[1, 0, 1, 0, 1, 0, 1], hidden=(64, 64), nnFunc="Tanh", batchSize=40, maxTotObsNum=3000,
          randSeed=12),
     dict(seed=13, dimS=32, dimA=7, lenMin=3, lenMax=9, pTerm=0.0), 120, 15),
    # batch 1024: 2000+ workgroups of the fused kernel, four times what the GPU holds at once -- the panel
    # groups must complete wave after wave (no group may wait for workgroups that cannot become resident)
    (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=1024, maxTotObsNum=100000, randSeed=2),
     dict(seed=19, dimS=17, dimA=6, lenMin=100, lenMax=200, pTerm=0.1), 200, 4),
    # run-time activation dispatch (neither SoftSign nor Tanh), 8 tiles per panel
    (dict(dimS=20, dimA=4, bounded=[0, 1, 1, 0], hidden=(128, 128), nnFunc="Relu", batchSize=100, maxTotObsNum=20000,
          randSeed=21, nnLambda=1e-5),
     dict(seed=17, dimS=20, dimA=4, lenMin=30, lenMax=90, pTerm=0.3), 150, 10),
    # BASELINE.json configs[0]: cart_pole_cpp + settings/VRACER.json with the code defaults of
    # Settings/HyperParameters.h:42-72 (observed state 5, one action bounded, 2x128 Tanh, batch 256,
    # C = sqrt(dA/2), replay 2^14 sqrt(dA+dS))
    (dict(dimS=5, dimA=1, bounded=[1], hidden=(128, 128), nnFunc="Tanh", batchSize=256, maxTotObsNum=40131,
          clipImpWeight=0.5 ** 0.5, gamma=0.995, learnrate=1e-4, explNoise=0.2 ** 0.5, randSeed=31),
     dict(seed=23, dimS=5, dimA=1, lenMin=20, lenMax=200, pTerm=0.9), 60, 10),
    # BASELINE.json configs[2]: Humanoid-v2 through apps/OpenAI_gym/HumanoidWrapper.py (257 observed states,
    # 17 unbounded actions, 2x256), the share of ONE of its 8 replicas: local batch 32 (generic five-launch path)
    (dict(dimS=257, dimA=17, bounded=[0] * 17, hidden=(256, 256), batchSize=32, maxTotObsNum=60000,
          clipImpWeight=(17 / 2.0) ** 0.5, randSeed=33),
     dict(seed=29, dimS=257, dimA=17, lenMin=20, lenMax=120, pTerm=0.5, muSpread=0.2), 40, 8),
    # RACER with the Gaussian advantage head (Math/Gaus_advantage.h): 6 actions -> 13 advantage outputs, 26 in all
    (dict(dimS=17, dimA=6, bounded=[1, 0, 1, 0, 1, 1], hidden=(64, 64), batchSize=64, maxTotObsNum=20000, randSeed=41,
          adv_kind=capi.ADV_GAUSSIAN),
     dict(seed=37, dimS=17, dimA=6, lenMin=20, lenMax=80, pTerm=0.4), 80, 12),
    # RACER with discrete actions (Discrete_policy / Discrete_advantage): 18 options (the full Atari action set), 37 outputs
    (dict(dimS=24, dimA=1, hidden=(128, 128), batchSize=64, maxTotObsNum=20000, randSeed=43,
          adv_kind=capi.ADV_DISCRETE, n_options=18),
     dict(seed=39, dimS=24, dimA=1, lenMin=20, lenMax=80, pTerm=0.4), 80, 12),
    # RACER on two LSTM layers (RACER_RNN.json family, BASELINE config 4): truncated BPTT over up to 16 steps, short and
    # long windows, truncated episode ends (t+1 forwarded through the recurrence)
    (dict(dimS=6, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=32, maxTotObsNum=20000, randSeed=47, gamma=0.99,
          adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1),
     dict(seed=41, dimS=6, dimA=1, lenMin=3, lenMax=60, pTerm=0.4), 80, 12),
    # LSTM layers of an odd width, short BPTT window, V-RACER head
    (dict(dimS=6, dimA=2, bounded=[1, 0], hidden=(24, 24), nnFunc="Tanh", batchSize=20, maxTotObsNum=20000, randSeed=57,
          nn_type=capi.NN_LSTM, nnBPTTseq=5),
     dict(seed=49, dimS=6, dimA=2, lenMin=3, lenMax=30, pTerm=0.4), 60, 10),
    # V-RACER on MGU layers (Layer_GRU.h; what a partially observable MDP gets with the default nnType), unequal widths
    (dict(dimS=7, dimA=2, bounded=[0, 1], hidden=(48, 32), nnFunc="Tanh", batchSize=24, maxTotObsNum=20000, randSeed=53,
          nn_type=capi.NN_MGU, nnBPTTseq=10),
     dict(seed=45, dimS=7, dimA=2, lenMin=3, lenMax=50, pTerm=0.4), 80, 12),
])
def test_device_sampler_and_update_match_oracle(hip_api, cfg_kw, sc_kw, n_eps, steps):
    """Device-side mt19937 sampler (Lemire + sort/unique/redraw), gather, MLP, head, ReF-ER
    bookkeeping and Adam versus the CPU oracle, step by step, with the library drawing its own
    indices.  Integer outputs bit-exact."""
    G, O = _pair(hip_api, cfg_kw, synth_cfg(**sc_kw), n_eps)
    for k in range(steps):
        G.step(1); O.step(1)
        _compare_step(G, O)
        sg, so = G.scalars(), O.scalars()
        assert abs(sg.beta - so.beta) <= 1e-12 * so.beta and sg.CmaxRet == so.CmaxRet
        assert sg.nFarPolicySteps == so.nFarPolicySteps, (k, sg.nFarPolicySteps, so.nFarPolicySteps)
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    wg, m1g, m2g = G.get_params(); wo, m1o, m2o = O.get_params()
    assert relinf(wg, wo) < TOL32 and relinf(m1g, m1o) < 2 * TOL32 and relinf(m2g, m2o) < 2 * TOL32
    # what the steps wrote back into the replay (V, advantage = Q - V, importance weights) and the Q statistics
    for field in (capi.EP_VALUE, capi.EP_ADVANTAGE, capi.EP_IMPW, capi.EP_DKL, capi.EP_DELTAQ):
        mg, mo = episode_arrays_by_tag(G, field), episode_arrays_by_tag(O, field)
        for tag in mo:
            assert np.allclose(mg[tag], mo[tag], rtol=1e-4, atol=1e-5), (field, tag)
    stg, sto = G.stats(), O.stats()
    for f in ("avgKLdivergence", "avgSquaredErr", "avgReturn", "avgQ", "stdevQ", "minQ", "maxQ"):
        assert np.isclose(getattr(stg, f), getattr(sto, f), rtol=1e-3, atol=1e-5), f


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps,n", [
    (dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42),
     dict(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5), 30, 64),
    # 256-step graph + 64 + 8 + 2 + eager remainder; short truncated episodes: the riders (sampler with
    # gather helpers, bookkeeping) handle next-state rows in every step
    (dict(dimS=32, dimA=7, bounded=[1, 0, 1, 0, 1, 0, 1], hidden=(64, 64), nnFunc="Tanh", batchSize=40, maxTotObsNum=3000,
          randSeed=12),
     dict(seed=13, dimS=32, dimA=7, lenMin=3, lenMax=9, pTerm=0.0), 120, 331),
    # batch 512 on the fused path: more workgroups than the GPU holds at once, riders included
    (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=512, maxTotObsNum=100000, randSeed=2),
     dict(seed=19, dimS=17, dimA=6, lenMin=100, lenMax=200, pTerm=0.1), 120, 12),
    # layout not served by the fused kernel: generic five-launch graph
    (dict(dimS=9, dimA=3, bounded=[0, 0, 0], hidden=(24, 16, 8), nnFunc="Tanh", batchSize=8, maxTotObsNum=1000, randSeed=5),
     dict(seed=3, dimS=9, dimA=3, lenMin=3, lenMax=30, pTerm=0.3), 20, 70),
])
def test_multi_step_graph_replay_matches_single_steps(hip_api, cfg_kw, sc_kw, n_eps, n):
    """hl_step(n) (hipGraph replay of the launch sequence) == n x hl_step(1) == oracle."""
    sc = synth_cfg(**sc_kw)
    G, O = _pair(hip_api, cfg_kw, sc, n_eps)
    G.step(n); O.step(n)
    _compare_step(G, O)
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < TOL32
    assert G.scalars().nGradSteps == n


@pytest.mark.parametrize("head", ["vracer", "racer_gaussian", "racer_discrete"])
def test_thousand_step_sweep_matches_oracle(hip_api, head):
    """Crosses step 1000: Episode::updateCumulative + whole-buffer Retrace + reward/state EMA -- for the three
    advantage heads (stored advantages are non-zero for the two RACER ones)."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42,
                  epsAnneal=5e-7)
    if head == "racer_gaussian":
        cfg_kw.update(adv_kind=capi.ADV_GAUSSIAN)
    if head == "racer_discrete":
        cfg_kw.update(adv_kind=capi.ADV_DISCRETE, n_options=6, dimA=1, bounded=[0])
    sc = synth_cfg(seed=7, dimS=5, dimA=cfg_kw["dimA"], lenMin=5, lenMax=40, pTerm=0.5)
    G, O = _pair(hip_api, cfg_kw, sc, 30)
    G.step(999); O.step(999)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    G.step(1); O.step(1)
    for field, tol in ((capi.EP_RETURN, 1e-4), (capi.EP_VALUE, 1e-4), (capi.EP_ADVANTAGE, 1e-4), (capi.EP_IMPW, 1e-4),
                       (capi.EP_DKL, 1e-4)):
        mg, mo = episode_arrays_by_tag(G, field), episode_arrays_by_tag(O, field)
        for tag in mo:
            assert np.allclose(mg[tag], mo[tag], rtol=tol, atol=tol), (field, tag)
    mG, sG, rG = G.get_scaling(); mO, sO, rO = O.get_scaling()
    assert np.allclose(np.concatenate([mG, sG, rG]), np.concatenate([mO, sO, rO]), rtol=1e-6, atol=1e-7)
    G.step(100); O.step(100)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert np.array_equal(G.readback(capi.TAP_FAR), O.readback(capi.TAP_FAR))
    sg, so = G.scalars(), O.scalars()
    assert abs(sg.beta - so.beta) <= 1e-9 * so.beta
    assert sg.nFarPolicySteps == so.nFarPolicySteps
    assert relinf(G.get_params()[0], O.get_params()[0]) < 1e-4
    stg, sto = G.stats(), O.stats()
    for f in ("avgKLdivergence", "avgSquaredErr", "avgReturn", "avgQ", "stdevQ", "minQ", "maxQ"):
        assert np.isclose(getattr(stg, f), getattr(sto, f), rtol=1e-3, atol=1e-5), f


VARIANTS = {
    "vracer": {},
    "racer_gaussian_lstm": dict(adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnFunc="Tanh", nnBPTTseq=6),
    "racer_discrete_mgu": dict(adv_kind=capi.ADV_DISCRETE, n_options=5, dimA=1, bounded=[0], nn_type=capi.NN_MGU, nnFunc="Tanh", nnBPTTseq=4),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_eviction_and_append_during_training(hip_api, variant, tmp_path):
    """FIFO removal (applyEpisodesRemovalAlgo, 'oldest') and episodes appended between steps; the cumulative_rewards.dat
    lines MemoryBuffer::pushBackEpisode writes for them (MemoryBuffer.cpp:481-507)."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 1], hidden=(32, 32), batchSize=16, maxTotObsNum=600, minTotObsNum=300,
                  randSeed=2)
    cfg_kw.update(VARIANTS[variant])
    sc = synth_cfg(seed=21, dimS=5, dimA=cfg_kw["dimA"], lenMin=10, lenMax=40, pTerm=0.4)
    G, O = _pair(hip_api, cfg_kw, sc, 24)
    G.set_episode_log(tmp_path / "g_rewards.dat"); O.set_episode_log(tmp_path / "o_rewards.dat")
    e = 24
    for k in range(30):
        G.step(1); O.step(1)
        _compare_step(G, O)
        for L in (G, O):
            L.append_episode(**synth_episode(sc, e, cfg_kw.get("n_options", 0)))
        e += 1
        sg, so = G.scalars(), O.scalars()
        assert sg.nStoredSteps == so.nStoredSteps and sg.nStoredEps == so.nStoredEps
    assert relinf(G.get_params()[0], O.get_params()[0]) < TOL32
    lg, lo = open(tmp_path / "g_rewards.dat").read().split("\n"), open(tmp_path / "o_rewards.dat").read().split("\n")
    assert len(lg) == len(lo) == 31 and lg == lo and lg[3].split()[0] == "4"      # "nGradSteps timeStamp agent nSteps totalReward"


@pytest.mark.parametrize("rule", ["farpolfrac", "maxkldiv", "minerror"])
def test_removal_rules_other_than_oldest(hip_api, rule):
    """ERoldSeqFilter (getERfilterAlgo, MemoryProcessing.cpp:261-298): the episode that leaves an over-full replay is the one with
    the most far-policy steps / the largest D_KL / the smallest TD error -- from the aggregates the device maintains.  Episodes
    keep arriving between steps; the slot ring, where such removals leave holes, wraps several times (compaction)."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 1], hidden=(32, 32), batchSize=16, maxTotObsNum=600, minTotObsNum=300, randSeed=2,
                  ERoldSeqFilter=rule)
    sc = synth_cfg(seed=21, dimS=5, dimA=2, lenMin=10, lenMax=40, pTerm=0.4)
    G, O = _pair(hip_api, cfg_kw, sc, 30)
    e = 30
    for k in range(700):
        for L in (G, O):
            L.append_episode(**synth_episode(sc, e))
        e += 1
        if k % 2 == 0:
            G.step(1); O.step(1)
            sg, so = G.scalars(), O.scalars()
            assert sg.nStoredSteps == so.nStoredSteps and sg.nStoredEps == so.nStoredEps, k
        if k % 50 == 0:
            _compare_step(G, O)
            tg = [G.episode_info(p)[0] for p in range(G.scalars().nStoredEps)]
            to = [O.episode_info(p)[0] for p in range(O.scalars().nStoredEps)]
            assert tg == to, k                                    # the same episodes survived, in the same order
    assert max(tg) - min(tg) > len(tg) + 10                       # (not first in, first out: older episodes are still there)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.parametrize("extra", [{}, dict(adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnFunc="Tanh", nnBPTTseq=6)])
def test_split_step_equals_fused_step(hip_api, extra):
    """hl_step_begin / exchanges / hl_step_end (host-side all-reduce hook) == hl_step (dense and recurrent layers)."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42, **extra)
    sc = synth_cfg(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5)
    A, _ = _pair(hip_api, cfg_kw, sc, 30)
    Bq = hip_learner(hip_api, capi.make_config(**cfg_kw))
    Bq.init_weights(); fill_synth(Bq, sc, 30); Bq.initialize()
    for _ in range(5):
        A.step(1)
        Bq.step_begin(); g = Bq.grad_fetch(); Bq.grad_store(g); Bq.step_end()
    assert np.array_equal(A.get_params()[0], Bq.get_params()[0])
    assert A.scalars().beta == Bq.scalars().beta


def _xchg_replicas(hip_api, cfg_kw, sc, connect):
    """Two HIP replicas on this GPU; connect=True: exchanging through each other's windows (one thread per replica, as the
    collectives wait for the peer), else to be driven through the host-exchange entry points."""
    import threading
    from oracle_api import synth_episode
    Ls = []
    for r in range(2):
        L = hip_learner(hip_api, capi.make_config(n_ranks=2, rank=r, **cfg_kw))
        L.init_weights()
        for e in range(r, 40, 2):
            L.append_episode(**synth_episode(sc, e, cfg_kw.get("n_options", 0)))
        Ls.append(L)
    if connect == "before":                   # the start-up statistics are then sums over both shards (hl_initialize exchanges them)
        handles = [L.xchg_export() for L in Ls]
        _both(Ls, lambda L: (L.xchg_connect(handles), L.initialize()))
        return Ls
    w0 = Ls[0].get_params()[0]
    for L in Ls:                              # host-exchange mode: statistics of the local shard
        w, m1, m2 = L.get_params(); L.set_params(w0, m1, m2); L.initialize()
    if connect == "after":
        handles = [L.xchg_export() for L in Ls]
        _both(Ls, lambda L: L.xchg_connect(handles))
    return Ls


def _both(Ls, fn):
    import threading
    errs = []

    def run(L):
        try:
            fn(L)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=run, args=(L,)) for L in Ls]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]


def _host_sum_step(Ls):
    for L in Ls:
        L.step_begin()
    g = np.sum([L.grad_fetch() for L in Ls], axis=0, dtype=np.float32)
    ms = [L.moments_fetch() for L in Ls]
    c = np.sum([L.counters_fetch() for L in Ls], axis=0)
    for L, m in zip(Ls, ms):
        L.grad_store(g)
        if m is not None:
            L.moments_store(np.sum(ms, axis=0))
        L.counters_store(c)
        L.step_end()


@pytest.mark.parametrize("extra", [{}, dict(hidden=(24, 16, 8), nnFunc="Tanh"), dict(hidden=(16, 16), nnFunc="Tanh", nn_type=capi.NN_LSTM, nnBPTTseq=4),
                                   dict(adv_kind=capi.ADV_DISCRETE, n_options=4, dimA=1, bounded=[0])],
                         ids=["fused-2x32", "generic-24x16x8", "lstm-2x16", "discrete-head"])
def test_one_kernel_exchange_between_two_replicas(hip_api, extra):
    """hl_xchg_export / hl_xchg_connect (xchg.hip): each replica writes its gradient-and-counters message straight into the
    other's window and sums in rank order -- against the same two replicas with the sums formed on the host: weights, Adam
    moments, beta and the generator bit for bit after eager calls, replayed graphs (the exchange kernel is a graph node; networks
    the fused kernel does not serve step eagerly) and the 1000th-step sweep with its moments exchange; both replicas identical
    throughout."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)
    cfg_kw.update(extra)
    sc = synth_cfg(seed=3, dimS=5, dimA=cfg_kw["dimA"], lenMin=8, lenMax=30, pTerm=0.5)
    X = _xchg_replicas(hip_api, cfg_kw, sc, "after")      # (same start as H: initialised on the local shards, then connected)
    H = _xchg_replicas(hip_api, cfg_kw, sc, None)
    assert X[0].B == 8
    done = 0
    for n in (1, 1, 3, 20, 70, 900, 10):
        _both(X, lambda L: (L.step(n), L.sync()))
        for _ in range(n):
            _host_sum_step(H)
        done += n
        for r in range(2):
            for a, b in zip(X[r].get_params(), H[r].get_params()):
                assert np.array_equal(a, b), (done, r)
            assert X[r].scalars().beta == H[r].scalars().beta and np.array_equal(X[r].get_rng_state(), H[r].get_rng_state())
        assert np.array_equal(X[0].get_params()[0], X[1].get_params()[0])
    coll = hip_api.lib.hl_debug_collectives
    coll.restype = C.c_int64; coll.argtypes = [C.c_void_p]
    assert coll(X[0].h) == coll(X[1].h) >= 1005


@pytest.mark.parametrize("n_ranks", [3, 4, 8])
def test_one_kernel_exchange_among_several_replicas(hip_api, n_ranks):
    """The exchange among 3, 4 and 8 replicas (one thread each, all on this GPU): every replica sums the contributions in rank order, so
    all end bit-identical; against host-formed sums in the same order -- ((g0 + g1) + g2) + g3 in fp32 -- over eager calls, replayed
    graphs and a 1000th-step sweep.  (Batch 24 splits 3 x 8 and 4 x 6.)"""
    from oracle_api import synth_episode
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=24, maxTotObsNum=6000, randSeed=11)
    sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)

    def replicas(connect):
        Ls = []
        for r in range(n_ranks):
            L = hip_learner(hip_api, capi.make_config(n_ranks=n_ranks, rank=r, **cfg_kw))
            L.init_weights()
            for e in range(r, 60, n_ranks):
                L.append_episode(**synth_episode(sc, e))
            Ls.append(L)
        w0 = Ls[0].get_params()[0]
        for L in Ls:
            w, m1, m2 = L.get_params(); L.set_params(w0, m1, m2); L.initialize()
        if connect:
            handles = [L.xchg_export() for L in Ls]
            _both(Ls, lambda L: L.xchg_connect(handles))
        return Ls

    def host_step(Ls):
        for L in Ls:
            L.step_begin()
        gs = [L.grad_fetch() for L in Ls]
        g = gs[0].copy()
        for q in gs[1:]:
            g = (g + q).astype(np.float32)                    # rank order, fp32
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

    X, H = replicas(True), replicas(False)
    assert X[0].B == 24 // n_ranks
    for n in (1, 2, 20, 70, 900, 12):
        _both(X, lambda L: (L.step(n), L.sync()))
        for _ in range(n):
            host_step(H)
        for r in range(n_ranks):
            for a, b in zip(X[r].get_params(), H[r].get_params()):
                assert np.array_equal(a, b), (n, r)
            assert X[r].scalars().beta == H[r].scalars().beta
        for r in range(1, n_ranks):
            assert np.array_equal(X[0].get_params()[0], X[r].get_params()[0])


def test_one_kernel_exchange_at_start_up(hip_api):
    """Connected BEFORE hl_initialize: rank 0's weights reach rank 1, the start-up counters and reward / state moments are summed
    over both shards (Learner::initializeLearner with several learners) -- the scaling equals that of ONE learner holding all
    episodes -- and the replicas stay identical over the following steps."""
    from oracle_api import synth_episode
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)
    sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)
    Y = _xchg_replicas(hip_api, cfg_kw, sc, "before")
    S = hip_learner(hip_api, capi.make_config(**cfg_kw)); S.init_weights()
    for e in range(40):
        S.append_episode(**synth_episode(sc, e))
    S.initialize()
    ref = np.concatenate(S.get_scaling())
    for r in range(2):
        got = np.concatenate(Y[r].get_scaling())
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-7), r
    assert all(np.array_equal(a, b) for a, b in zip(Y[0].get_scaling(), Y[1].get_scaling()))
    assert np.array_equal(Y[0].get_params()[0], Y[1].get_params()[0])
    c0, c1 = Y[0].counts(), Y[1].counts()
    _both(Y, lambda L: (L.step(30), L.sync()))
    assert np.array_equal(Y[0].get_params()[0], Y[1].get_params()[0]) and Y[0].scalars().beta == Y[1].scalars().beta


@pytest.mark.parametrize("n_procs,shape", [(2, "small"), (8, "small"), (2, "north-star"), (8, "north-star")], ids=["2", "8", "2-north-star", "8-north-star"])
def test_one_kernel_exchange_between_processes(n_procs, shape):
    """The same exchange between 2 and 8 PROCESSES sharing this GPU -- the layout of a node's eight learner ranks, minus the
    links --, windows mapped through hipIpc handles that travel over gloo (tests/xchg_ipc_worker.py): replicas identical and equal
    to the host-summed run (rank-order fp32 sums), bit for bit, after 1005 steps; round 6: also at BASELINE.json's metric
    configuration (2 x 256, global batch 256: the 292 KB message pushed by 354 dW tiles into hipIpc-mapped windows)."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SMARTIES_HIP_XCHG_TIMEOUT_MS="60000", XCHG_IPC_SHAPE=shape)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_procs, "--master-addr", "127.0.0.1",
                          "--master-port", str(29531 + n_procs + (40 if shape != "small" else 0)), os.path.join(here, "xchg_ipc_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert "XCHG_IPC_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_exchange_times_out_instead_of_hanging(hip_api):
    """A peer that never issues its collective: the waiting replica's kernel gives up after SMARTIES_HIP_XCHG_TIMEOUT_MS and the
    learner reports a device error; the GPU stays usable."""
    import subprocess, sys, os
    code = (
        "import os, sys; sys.path[:0] = [%r, %r]\n"
        "from smarties_amd import capi, load_hip; from oracle_api import synth_cfg, synth_episode\n"
        "api = load_hip(); sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)\n"
        "Ls = [capi.Learner(api, capi.make_config(n_ranks=2, rank=r, dimS=5, dimA=2, hidden=(32, 32), batchSize=16)) for r in range(2)]\n"
        "hs = [L.xchg_export() for L in Ls]\n"
        "try:\n    Ls[0].xchg_connect(hs); print('NO_ERROR')\n"
        "except capi.HlError as e:\n    print('TIMED_OUT', e)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMARTIES_HIP_XCHG_TIMEOUT_MS="200"), capture_output=True, text=True, timeout=300)
    assert "TIMED_OUT" in out.stdout, out.stdout + out.stderr[-2000:]
    L = hip_learner(hip_api, capi.make_config(dimS=5, dimA=2, hidden=(16,), batchSize=8))     # the device still works
    L.init_weights()


@pytest.mark.gpu
def test_panel_exchange_of_the_fused_kernel_checks_where_its_workgroups_run(hip_api):
    """The fused kernel hands y3 / f'(x2) between the workgroups of a panel with plain stores and loads, sound only while they share an
    XCD's L2.  hl_create probes HW_REG_XCC_ID with the kernel's launch shape; if workgroup b is not on XCD b % 8 -- or with
    SMARTIES_HIP_PANEL_SAFE=1 -- the exchange goes through agent-scope accesses.  Same bits either way."""
    import subprocess, sys, os
    mode = hip_api.lib.hl_debug_panel_mode; mode.restype = C.c_int; mode.argtypes = [C.c_void_p]
    L = hip_learner(hip_api, capi.make_config(dimS=5, dimA=2, hidden=(64, 64), batchSize=64, maxTotObsNum=4000))
    assert mode(L.h) == 0, "the dispatcher no longer deals workgroup b to XCD b % 8 on this system (the library then takes the safe path)"
    code = (
        "import os, sys, hashlib, ctypes as C; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from smarties_amd import capi, load_hip; from oracle_api import synth_cfg, fill_synth\n"
        "api = load_hip(); sc = synth_cfg(seed=3, dimS=17, dimA=6, lenMin=20, lenMax=60, pTerm=0.3)\n"
        "L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=40000))\n"
        "L.init_weights(); fill_synth(L, sc, 300); L.initialize(); L.step(1); L.step(300); L.sync()\n"
        "m = api.lib.hl_debug_panel_mode; m.restype = C.c_int; m.argtypes = [C.c_void_p]\n"
        "print('MODE', m(L.h), 'HASH', hashlib.sha1(b''.join(a.tobytes() for a in L.get_params())).hexdigest(), L.scalars().beta)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for safe in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMARTIES_HIP_PANEL_SAFE=safe), capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("MODE")]
        assert line, out.stdout + out.stderr[-2000:]
        outs.append(line[-1].split())
    assert outs[0][1] == "0" and outs[1][1] == "1"
    assert outs[0][3:] == outs[1][3:], outs


@pytest.mark.gpu
def test_a_missing_peer_leaves_the_learner_state_intact():
    """Two connected replicas step together, then only one issues a step: its exchange kernel gives up after the (shortened) wait,
    raises the sticky device error -- and applies NOTHING of that step: no sum over stale slots, no Adam, no bookkeeping.  Weights,
    moments and beta are those of the last completed step."""
    import subprocess, sys, os
    code = (
        "import os, sys, threading; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from smarties_amd import capi, load_hip; from oracle_api import synth_cfg, synth_episode\n"
        "api = load_hip(); sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)\n"
        "Ls = []\n"
        "for r in range(2):\n"
        "    L = capi.Learner(api, capi.make_config(n_ranks=2, rank=r, dimS=5, dimA=2, hidden=(32, 32), batchSize=16, maxTotObsNum=4096))\n"
        "    L.init_weights()\n"
        "    for e in range(r, 40, 2): L.append_episode(**synth_episode(sc, e))\n"
        "    Ls.append(L)\n"
        "hs = [L.xchg_export() for L in Ls]\n"
        "def both(fn):\n"
        "    ts = [threading.Thread(target=fn, args=(L,)) for L in Ls]\n"
        "    [t.start() for t in ts]; [t.join() for t in ts]\n"
        "both(lambda L: (L.xchg_connect(hs), L.initialize()))\n"
        "both(lambda L: (L.step(5), L.sync()))\n"
        "before = [a.copy() for a in Ls[0].get_params()]; beta = Ls[0].scalars().beta\n"
        "Ls[0].step(1)\n"                                     # replica 1 never comes
        "try:\n    Ls[0].scalars(); print('NO_ERROR')\n"
        "except capi.HlError as e:\n    print('TIMED_OUT', e)\n"
        "after = Ls[0].get_params()\n"
        "print('INTACT' if all(np.array_equal(a, b) for a, b in zip(before, after)) else 'CHANGED')\n"
        "sys.stdout.flush(); os._exit(0)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMARTIES_HIP_XCHG_TIMEOUT_MS="300", GPU_MAX_HW_QUEUES="8"),
                         capture_output=True, text=True, timeout=300)
    assert "TIMED_OUT" in out.stdout and "INTACT" in out.stdout, out.stdout + out.stderr[-2000:]


@pytest.mark.gpu
def test_two_replica_protocol_matches_oracle_replicas(hip_api):
    """n_ranks = 2 (batch and replay budget split): two HIP replicas on this GPU,
    exchanges summed on the host, against two oracle replicas driven the same way -- over a
    1000th-step sweep so that the moments exchange is part of it."""
    from oracle_api import oracle_learner, synth_episode
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)
    sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)

    def replicas(make):
        Ls = []
        for r in range(2):
            L = make(capi.make_config(n_ranks=2, rank=r, **cfg_kw))
            L.init_weights()
            for e in range(r, 40, 2):
                L.append_episode(**synth_episode(sc, e))
            Ls.append(L)
        w0 = Ls[0].get_params()[0]
        for L in Ls:
            w, m1, m2 = L.get_params(); L.set_params(w0, m1, m2); L.initialize_begin()
        c = np.sum([L.counters_fetch() for L in Ls], axis=0)      # the start-up statistics are global (Learner.cpp:58-59: accurate reductions)
        m = np.sum([L.moments_fetch() for L in Ls], axis=0)
        for L in Ls:
            L.counters_store(c); L.moments_store(m); L.initialize_end()
        return Ls

    def one_step(Ls):
        for L in Ls:
            L.step_begin()
        gs = [L.grad_fetch() for L in Ls]
        g = np.sum(gs, axis=0, dtype=np.float32)
        ms = [L.moments_fetch() for L in Ls]
        c = np.sum([L.counters_fetch() for L in Ls], axis=0)
        for L, m in zip(Ls, ms):
            L.grad_store(g)
            if m is not None:
                L.moments_store(np.sum(ms, axis=0))
            L.counters_store(c)
            L.step_end()
        return gs, c, ms

    G = replicas(lambda cfg: hip_learner(hip_api, cfg))
    O = replicas(oracle_learner)
    assert G[0].B == 8 and O[0].B == 8
    for k in range(1, 1004):
        for L in G + O:
            L.set_tap(k <= 3 or k >= 999)
        gG, cG, mG = one_step(G)
        gO, cO, mO = one_step(O)
        if k <= 3 or k >= 999:
            for r in range(2):
                assert np.array_equal(G[r].readback(capi.TAP_FLAT), O[r].readback(capi.TAP_FLAT)), (k, r)
                den = np.abs(gO[r]).max() + 1e-30
                assert np.abs(gG[r] - gO[r]).max() / den < 1e-5, (k, r)     # north_star: 1e-5 rel, fp32
            assert np.array_equal(cG[:2], cO[:2]) and cG[3] == cO[3]
            assert int(cG[2]) == int(cO[2])                                 # far-policy count
        if k == 1000:
            assert mG[0] is not None and mO[0] is not None
            assert np.allclose(mG[0], mO[0], rtol=1e-12, atol=1e-9)
    for r in range(2):
        wG, wO = G[r].get_params()[0], O[r].get_params()[0]
        assert np.abs(wG - wO).max() < 2e-4
        assert abs(G[r].scalars().beta - O[r].scalars().beta) < 1e-3
    assert np.array_equal(G[0].get_params()[0], G[1].get_params()[0])       # replicas stay identical


@pytest.mark.gpu
def test_rccl_exchange_sequence_on_one_rank(hip_api):
    """hl_comm_init on a single replica switches hl_step to the N > 1 sequence (gradient ->
    ncclAllReduce -> separate Adam kernel -> counters ncclAllReduce -> beta update, moments
    all-reduce on the 1000th step) over a 1-rank RCCL communicator: same results as the fused
    single-replica step."""
    import ctypes as C
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42)
    sc = synth_cfg(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5)
    A, _ = _pair(hip_api, cfg_kw, sc, 30)
    Bq = hip_learner(hip_api, capi.make_config(**cfg_kw))
    Bq.init_weights(); fill_synth(Bq, sc, 30)
    raw = (C.c_uint8 * 128)()
    assert hip_api.fn("comm_unique_id")(raw) == 0
    Bq.comm_init(bytes(raw))
    Bq.initialize()
    for n in (1, 7, 64, 931):                # 1003 steps: crosses the periodic sweep
        A.step(n); Bq.step(n)
        assert np.array_equal(A.get_params()[0], Bq.get_params()[0]), n
        assert A.scalars().beta == Bq.scalars().beta
        assert A.scalars().nFarPolicySteps == Bq.scalars().nFarPolicySteps
    assert np.array_equal(A.get_rng_state(), Bq.get_rng_state())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "racer_lstm.bin", "vracer_mgu.bin", "racer_discrete.bin"])
def test_checkpoint_files_match_reference(hip_api, name, tmp_path):
    """hl_save == the reference's own checkpoint files (Network::save, Network.cpp:22-38) for the
    same weights / moments; hl_restart of the reference's files restores the device blobs; a
    truncated file is refused."""
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    L.set_params(fx["Wfinal"], fx["M1final"], fx["M2final"])
    base = str(tmp_path / "agent_00_net")
    L.save(base)
    for suf in ("_weights", "_1stMom", "_2ndMom"):
        assert open(base + suf + ".raw", "rb").read() == bytes(bytearray(fx["ckpt_net" + suf])), suf
    L2 = hip_learner(hip_api, fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    L2.init_weights()
    L2.restart(base)
    w, m1, m2 = L2.get_params()
    assert np.array_equal(w, fx["Wfinal"]) and np.array_equal(m1, fx["M1final"]) and np.array_equal(m2, fx["M2final"])
    data = open(base + "_weights.raw", "rb").read()
    open(base + "_weights.raw", "wb").write(data[:-8])
    with pytest.raises(capi.HlError) as e:
        L2.restart(base)
    assert e.value.status == 7                      # HL_ERR_IO


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_mixed.bin", "racer_discrete.bin"])
def test_packed_episode_wire_format_matches_reference(hip_api, name):
    from parity import check_packed_roundtrip
    check_packed_roundtrip(lambda cfg: hip_learner(hip_api, cfg), load_fixture(name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_mixed.bin", "racer_discrete.bin"])
def test_memory_checkpoint_files_of_the_reference(hip_api, tmp_path, name):
    """hl_restart_memory reads the replay-memory checkpoint the COMPILED REFERENCE wrote after 12 gradient
    steps (MemoryBuffer::save, MemoryBuffer.cpp:274-324): scaling, counters, ReF-ER state and every per-step
    field of every episode come back exactly; hl_save_memory of that state reproduces the three files byte
    for byte."""
    fx = load_fixture(name)
    dA, polDim = int(fx["cfg"][1]), (int(fx["cfg"][12]) if len(fx["cfg"]) > 12 and fx["cfg"][12] else 2 * int(fx["cfg"][1]))
    base = str(tmp_path / "agent_00")
    names = ("_scaling", "_rank_000_learner_status", "_rank_000_learner_data")
    for suf in names:
        open(base + suf + ".raw", "wb").write(bytes(bytearray(fx["memck" + suf])))
    L = hip_learner(hip_api, fixture_config(fx))
    L.init_weights()
    L.restart_memory(base)
    s = L.scalars()
    assert (s.nStoredEps, s.nStoredSteps, s.nGradSteps) == (30, 659, 13)
    assert s.CmaxRet == 5.0 and abs(s.beta - 1.044926e-02) < 1e-8
    scal = np.frombuffer(bytes(bytearray(fx["memck_scaling"])), np.float64)
    m, sc_, r3 = L.get_scaling()
    assert np.array_equal(m, scal[0:5].astype(np.float32)) and np.array_equal(sc_, scal[5:10].astype(np.float32))
    assert np.array_equal(np.asarray(r3, np.float32), scal[[17, 16, 15]].astype(np.float32))    # file: std, scale, mean
    data = bytes(bytearray(fx["memck_rank_000_learner_data"]))
    off = 0
    for i in range(30):                                  # file order = oldest first = position 29 - i
        n = int(np.frombuffer(data[off:off + 8], np.uint64)[0]); off += 8
        size = (5 + 1 + dA + polDim + 6) * n + 10
        rec = data[off:off + 4 * size]; off += 4 * size
        assert L.pack_episode(29 - i).tobytes() == rec, i
    assert off == len(data)
    out = str(tmp_path / "copy")
    L.save_memory(out)
    for suf in names:
        mine, ref = open(out + suf + ".raw", "rb").read(), bytes(bytearray(fx["memck" + suf]))
        if suf.endswith("status"):
            # the reference writes counters.nGradSteps + 1 and restarts into counters.nGradSteps without
            # taking the 1 back (MemoryBuffer.cpp:303, 250-251): every save / restart cycle adds one
            assert mine == ref.replace(b"nGradSteps: 13", b"nGradSteps: 14")
        else:
            assert mine == ref, suf
    L.step(5)                                            # restarted learners step without initializeLearner
    assert L.scalars().nGradSteps == 18


@pytest.mark.gpu
def test_stats_line_matches_reference_log(hip_api):
    """Replay the 12 steps the reference took (same sampled pairs), then print the statistics line: same
    header and -- number by number, at the printed precision -- the same line as the reference's own
    Learner::logStats wrote for that state (MemoryBuffer::getMetrics + AdamOptimizer::getMetrics)."""
    fx = load_fixture("small_mixed.bin")
    L = hip_learner(hip_api, fixture_config(fx))
    setup_from_fixture(L, fx)
    for k in range(1, int(fx["cfg"][4]) + 1):
        L.step(1, flat=np.sort(our_flat_for(L, fx["s%d_tag" % k], fx["s%d_t" % k]), kind="stable"))
    head, line = L.metrics()
    ref_head = bytes(bytearray(fx["metrics_head"])).decode()
    assert head == ref_head
    assert lines_agree(line, bytes(bytearray(fx["metrics_line"])).decode(), head), line
    assert lines_agree(line, stats_line(L), head)


@pytest.mark.gpu
def test_resumed_run_follows_the_reference_after_its_own_restart(hip_api, tmp_path):
    """Resume as the reference resumes (Learner_approximator::restart, Learner_approximator.cpp:118-131): one run of the compiled
    reference trained 40 steps and wrote its network and replay-memory checkpoints (resume_first.bin carries the files), a SECOND
    reference process restarted from them -- skipping initializeLearner, Learner.cpp:51-54 -- and trained 20 more steps with taps
    (resume_second.bin).  The library restarted from the same files follows that second run: generator state before its first step,
    outputs, importance weights, gradients, weights and Adam moments, beta.  (What the checkpoint does not hold restarts as in the
    reference: Adam's running bias-correction factors begin again at beta_1, beta_2 while the step count continues.)"""
    fa, fr = load_fixture("resume_first.bin"), load_fixture("resume_second.bin")
    base = str(tmp_path / "ck")
    for suf in ("_net_weights", "_net_1stMom", "_net_2ndMom"):
        open(base + suf + ".raw", "wb").write(bytes(bytearray(fa["ckpt" + suf])))
    for suf in ("_scaling", "_rank_000_learner_status", "_rank_000_learner_data"):
        open(base + suf + ".raw", "wb").write(bytes(bytearray(fa["memck" + suf])))
    L = hip_learner(hip_api, fixture_config(fr))
    L.init_weights()                                  # (the restarted process builds and initialises its network first, too)
    L.restart(base + "_net"); L.restart_memory(base)
    assert np.array_equal(L.get_params()[0], fa["Wfinal"])
    assert L.scalars().nGradSteps == 41 and L.scalars().nStoredSteps == int(fr["cfg"][7])     # (MemoryBuffer::save writes nGradSteps + 1)
    assert np.array_equal(L.get_rng_state(), fr["s1_rng"])
    assert abs(L.scalars().beta - fr["s1_beta"][0]) <= 1e-6 * fr["s1_beta"][0]       # (the status file holds 7 digits)
    # the harness names episodes by agentID; restored episodes keep it in their wire record (trailer: terminated flag, ID, sampled
    # count, agentID -- Episode.cpp:78-81), from which the (episode, t) pairs of the reference are mapped to this replay's indices
    prefix, acc = {}, 0
    for p in range(L.scalars().nStoredEps):
        rec = L.pack_episode(p).tobytes()
        prefix[int.from_bytes(rec[-40:][17:25], "little", signed=True)] = acc
        acc += L.episode_info(p)[1] - 1
    assert len(prefix) == 30
    for k in range(1, int(fr["cfg"][4]) + 1):
        sk = "s%d_" % k
        flat = np.array([prefix[int(g)] + int(t) for g, t in zip(fr[sk + "tag"], fr[sk + "t"])], np.int64)
        order = np.argsort(flat, kind="stable")
        L.step(1, flat=flat[order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fr[sk + "O"][order]) < TOL32, k
        assert relinf(L.readback(capi.TAP_RHO), fr[sk + "rho"][order]) < TOL32, k
        assert relinf(L.readback(capi.TAP_OUTGRAD), fr[sk + "G"][order]) < TOL32, k
        assert np.array_equal(L.readback(capi.TAP_FAR), fr[sk + "far"][order]), k
        if sk + "W" in fr:
            w, m1, m2 = L.get_params()
            assert relinf(w, fr[sk + "W"]) < TOL32 and relinf(m1, fr[sk + "M1"]) < TOL32 and relinf(m2, fr[sk + "M2"]) < 2 * TOL32, k
        assert abs(L.scalars().beta - fr["traj_beta"][k - 1]) <= 2e-6 * fr["traj_beta"][k - 1], k
    assert relinf(L.get_params()[0], fr["Wfinal"]) < TOL32


@pytest.mark.gpu
def test_episode_log_equals_the_file_the_reference_wrote(hip_api, tmp_path):
    """cumulative_rewards.dat (MemoryBuffer::pushBackEpisode, MemoryBuffer.cpp:491-513): the file the compiled reference wrote while
    moving_replay.bin was recorded -- 25 episodes before training, 30 more behind every second gradient step -- against the
    library's hl_set_episode_log for the same arrivals: gradient-step count, time stamp (observations since minTotObsNum), length
    and total reward of every line.  (The harness logs its content tag as the agent id; the library's callers log agent 0.)"""
    fx = load_fixture("moving_replay.bin")
    L = hip_learner(hip_api, fixture_config(fx))
    L.set_episode_log(tmp_path / "rewards.dat")
    setup_from_fixture(L, fx)
    for k in range(1, int(fx["cfg"][4]) + 1):
        L.step(1)
        e = fixture_arrival(fx, k)
        if e is not None:
            L.append_episode(**synth_episode(fixture_synth(fx), e))
    L.sync()
    ref = bytes(fx["rewards_log"]).decode().splitlines()
    mine = open(tmp_path / "rewards.dat").read().splitlines()
    assert len(ref) == len(mine) == 55
    for a, b in zip(ref, mine):
        a, b = a.split(), b.split()
        assert [a[0], a[1], a[3], a[4]] == [b[0], b[1], b[3], b[4]], (a, b)


@pytest.mark.gpu
def test_stats_line_with_episodes_arriving_between_steps(hip_api):
    """totEp / totObs of the statistics line are the seen counters AS OF THE LAST updateCounters (ReplayCounters::nSeenEpisodes,
    MemoryProcessing.cpp:60-61), not this instant's: episodes appended since the last step do not show yet (nEp / nObs, the stored
    ones, do).  The oracle prints the compiled reference's line in that situation (moving_traj_1200.bin); the library prints the
    oracle's."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=1200, minTotObsNum=400, randSeed=42)
    sc = synth_cfg(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5)
    G, O = _pair(hip_api, cfg_kw, sc, 30)
    e = 30
    for k in range(1, 43):
        G.step(1); O.step(1)
        if k % 3 == 0:
            for L in (G, O):
                L.append_episode(**synth_episode(sc, e))
            e += 1
    head, line = G.metrics()
    assert lines_agree(line, stats_line(O), head, rel=2e-5), (line, stats_line(O))
    seen = int(line.split()[head.replace("|", " ").split().index("totEp")])
    assert seen == e - 1 == G.counts()[4] - 1                    # the episode appended behind step 39 shows, the one behind step 42 not yet
    assert G.scalars().nSeenEps == O.scalars().nSeenEps == seen


@pytest.mark.gpu
def test_output_gradient_statistics_file_matches_reference(hip_api, tmp_path):
    """StatsTracker (Utils/StatsTracker.cpp): <learner>_net_outGrad_stats.raw as the compiled reference wrote
    it for the same 12 steps (one record, at step 0), and the mean / RMS over the last minibatch."""
    fx = load_fixture("small_mixed.bin")
    L = hip_learner(hip_api, fixture_config(fx))
    setup_from_fixture(L, fx)
    L.set_log_base(str(tmp_path / "agent_00"))
    for k in range(1, int(fx["cfg"][4]) + 1):
        L.step(1, flat=np.sort(our_flat_for(L, fx["s%d_tag" % k], fx["s%d_t" % k]), kind="stable"))
    ref_file = np.frombuffer(bytes(bytearray(fx["outgrad_stats_file"])), np.float32)
    mine_file = np.fromfile(str(tmp_path / "agent_00_net_outGrad_stats.raw"), np.float32)
    assert mine_file.size == ref_file.size == 1 + 2 * L.nOut and mine_file[0] == ref_file[0]
    assert relinf(mine_file[1:], ref_file[1:]) < TOL32
    m, r = L.grad_stats()
    assert relinf(np.concatenate([m, r]), fx["outgrad_stats_last"]) < TOL32


@pytest.mark.gpu
def test_output_gradient_statistics_across_graph_replays_match_oracle(hip_api, tmp_path):
    """2100 free-running steps with the log switched on: records at steps 0, 1000 and 2000 (the library has to
    leave its graphs for exactly those steps), equal to the oracle's file."""
    fx = load_fixture("traj_1200.bin")
    G, O = hip_learner(hip_api, fixture_config(fx)), oracle_learner(fixture_config(fx))
    for L, nm in ((G, "g"), (O, "o")):
        setup_from_fixture(L, fx)
        L.set_log_base(str(tmp_path / nm))
        L.step(700); L.step(1400)
    fg = np.fromfile(str(tmp_path / "g_net_outGrad_stats.raw"), np.float32)
    fo = np.fromfile(str(tmp_path / "o_net_outGrad_stats.raw"), np.float32)
    assert fg.size == fo.size == 1 + 3 * 2 * G.nOut and fg[0] == fo[0]
    assert np.allclose(fg, fo, rtol=1e-3, atol=1e-6)
    assert G.scalars().nGradSteps == O.scalars().nGradSteps == 2100
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))


@pytest.mark.gpu
def test_stats_line_after_the_thousand_step_sweep_matches_oracle(hip_api):
    """1200 free-running steps of the traj_1200 configuration (larger numbers: other precision branches of
    real2SS, Q statistics present, beta column): the library's line against the line rebuilt from the
    oracle's state (tests/parity.stats_line, itself pinned to the reference's line in the CPU suite)."""
    fx = load_fixture("traj_1200.bin")
    G, O = hip_learner(hip_api, fixture_config(fx)), oracle_learner(fixture_config(fx))
    for L in (G, O):
        setup_from_fixture(L, fx)
        L.step(1000)
    (hg, lg), (ho, lo) = G.metrics(), O.metrics()      # the line of step 1000 (Learner::logStats): with the dRet column of the sweep
    assert hg == ho and "dRet" in hg and lines_agree(lg, lo, hg, rel=1e-3), (lg, lo)
    for L in (G, O):
        L.step(200)
    head, line = G.metrics()
    assert head == bytes(bytearray(fx["metrics_head"])).decode()
    assert lines_agree(line, stats_line(O), head, rel=1e-3), (line, stats_line(O))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_memory_and_network_checkpoint_round_trip_continues_identically(hip_api, tmp_path, variant):
    """Save network + memory after 40 steps, restart both into a fresh learner (plus the generator
    state, which the reference does not checkpoint), continue both for 30 steps: identical."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=2000, randSeed=42)
    cfg_kw.update(VARIANTS[variant])
    sc = synth_cfg(seed=7, dimS=5, dimA=cfg_kw["dimA"], lenMin=5, lenMax=40, pTerm=0.5)
    A = hip_learner(hip_api, capi.make_config(**cfg_kw))
    A.init_weights(); fill_synth(A, sc, 30); A.initialize(); A.step(40)
    base = str(tmp_path / "agent_00")
    A.save(base + "_net"); A.save_memory(base)
    Bq = hip_learner(hip_api, capi.make_config(**cfg_kw))
    Bq.init_weights(); Bq.restart(base + "_net"); Bq.restart_memory(base)
    Bq.set_rng_state(A.get_rng_state())
    for p in range(30):
        assert A.pack_episode(p).tobytes() == Bq.pack_episode(p).tobytes()      # (packing rounds to fp32 on both sides)
    # the status file is text: beta and CmaxReFER carry 7 significant digits (%le), as in the reference
    assert abs(Bq.scalars().beta - A.scalars().beta) <= 1e-6 * A.scalars().beta
    A.step(1); Bq.step(1)
    assert np.array_equal(A.readback(capi.TAP_FLAT), Bq.readback(capi.TAP_FLAT))
    assert np.array_equal(A.readback(capi.TAP_OUTPUT), Bq.readback(capi.TAP_OUTPUT))
    # actions and behaviour policies travel as fp32 in the episode records (Episode.cpp:38-40)
    assert relinf(A.readback(capi.TAP_RHO), Bq.readback(capi.TAP_RHO)) < 1e-5


@pytest.mark.gpu
def test_long_replayed_runs_are_deterministic(hip_api):
    """Two learners, same seed, 6000 steps each through the replayed graphs (sampler and gather riders,
    in-kernel panel barriers, bookkeeping rider, 1000-step sweeps): weights, moments, beta, far-policy count
    and generator state bit-identical, no device-side wait ever timed out (get_scalars would raise)."""
    cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=200000, randSeed=3)
    sc = synth_cfg(seed=23, dimS=17, dimA=6, lenMin=100, lenMax=200, pTerm=0.2)
    out = []
    for run in range(2):
        L = hip_learner(hip_api, capi.make_config(**cfg_kw))
        L.init_weights(); fill_synth(L, sc, 400); L.initialize()
        L.step(6000)
        w, m1, m2 = L.get_params(); s = L.scalars()
        out.append((w, m1, m2, s.beta, s.nFarPolicySteps, L.get_rng_state()))
        assert np.isfinite(w).all() and s.nGradSteps == 6000
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_sampler_collisions_and_redraw_match_oracle(hip_api):
    """Replay barely larger than the batch: most draws collide, so Sample_uniform's
    sort / unique / redraw-the-tail loop (Sampling.cpp:75-93) runs several rounds per step.  Indices,
    generator state and updates stay bit-exact / within tolerance; episodes of the minimal length
    (2 states) and terminated ones are part of the mix."""
    cfg_kw = dict(dimS=4, dimA=2, bounded=[1, 1], hidden=(16, 16), batchSize=32, maxTotObsNum=200, randSeed=77)
    sc = synth_cfg(seed=5, dimS=4, dimA=2, lenMin=2, lenMax=5, pTerm=0.5)
    G, O = _pair(hip_api, cfg_kw, sc, 16)
    nT = G.scalars().nStoredSteps
    assert 32 <= nT <= 64, nT                      # > 25 % duplicates expected per draw of 32
    for k in range(25):
        G.step(1); O.step(1)
        _compare_step(G, O)
        assert np.array_equal(G.get_rng_state(), O.get_rng_state()), k
    G.step(70); O.step(70)                         # the same through replayed graphs
    _compare_step(G, O)
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps", [
    # every episode is the shortest possible one (two states = one transition, all truncated: every sample has a next row)
    (dict(dimS=3, dimA=1, bounded=[0], hidden=(16, 16), batchSize=8, maxTotObsNum=100, randSeed=3),
     dict(seed=61, dimS=3, dimA=1, lenMin=2, lenMax=2, pTerm=0.0), 40),
    # the replay holds exactly one minibatch: the sampler has to return every transition, each step
    (dict(dimS=4, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=24, maxTotObsNum=100, randSeed=4),
     dict(seed=62, dimS=4, dimA=2, lenMin=4, lenMax=4, pTerm=1.0), 8),
    # batch of one (settings allow it; HyperParameters.cpp:186 leaves batchSize 1 unsplit), ragged episode lengths
    (dict(dimS=4, dimA=2, bounded=[1, 1], hidden=(16, 16), batchSize=1, maxTotObsNum=500, randSeed=5),
     dict(seed=63, dimS=4, dimA=2, lenMin=2, lenMax=17, pTerm=0.5), 12),
])
def test_edge_shapes_match_oracle(hip_api, cfg_kw, sc_kw, n_eps):
    G, O = _pair(hip_api, cfg_kw, synth_cfg(**sc_kw), n_eps)
    for k in range(6):
        G.step(1); O.step(1)
        _compare_step(G, O)
        assert G.scalars().beta == pytest.approx(O.scalars().beta, rel=1e-12)
    G.step(20); O.step(20)                                   # replayed graphs
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    if cfg_kw["batchSize"] == 24:
        assert np.array_equal(G.readback(capi.TAP_FLAT), np.arange(24))


def _random_config(rng):
    """A small learner configuration drawn at random: layer type, head, widths, batch, state / action sizes."""
    nn = [capi.NN_FFNN, capi.NN_FFNN, capi.NN_LSTM, capi.NN_MGU][int(rng.integers(4))]
    head = [capi.ADV_ZERO, capi.ADV_GAUSSIAN, capi.ADV_DISCRETE][int(rng.integers(3))]
    n_layers = int(rng.integers(1, 4))
    if nn == capi.NN_FFNN:
        hidden = [int(rng.integers(5, 72)) for _ in range(n_layers)]
    else:                                   # recurrent layers behind a residual must not widen (hl_create refuses that)
        w0 = 8 * int(rng.integers(1, 8))
        hidden = [w0] + [8 * int(rng.integers(1, w0 // 8 + 1)) for _ in range(n_layers - 1)]
        hidden = sorted(hidden, reverse=True)
    dimS = int(rng.integers(1, 40))
    kw = dict(dimS=dimS, hidden=tuple(hidden), nnFunc=["SoftSign", "Tanh", "Relu"][int(rng.integers(3))], nn_type=nn,
              batchSize=int(rng.integers(1, 70)), maxTotObsNum=4000, randSeed=int(rng.integers(1, 1000)), adv_kind=head,
              nnBPTTseq=int(rng.integers(1, 12)), nnLambda=float(rng.choice([0.0, 1e-5])),
              clipImpWeight=float(rng.choice([0.7, 2.0, 4.0])), gamma=float(rng.choice([0.9, 0.995])))
    if head == capi.ADV_DISCRETE:
        kw.update(dimA=1, bounded=[0], n_options=int(rng.integers(2, 20)))
    else:
        dA = int(rng.integers(1, 8 if head == capi.ADV_ZERO else 6))
        kw.update(dimA=dA, bounded=[int(b) for b in rng.integers(0, 2, dA)])
    sc = dict(seed=int(rng.integers(1, 1000)), dimS=dimS, dimA=kw["dimA"], lenMin=2, lenMax=int(rng.integers(3, 40)),
              pTerm=float(rng.choice([0.0, 0.5, 1.0])))
    return kw, sc


@pytest.mark.gpu
@pytest.mark.parametrize("func", ["LRelu", "Sigm", "HardSign", "SoftPlus", "ExpPlus", "Exp"])
def test_other_activation_functions_on_the_generic_path(hip_api, func):
    """The six remaining names of makeFunction (Functions.h:643-668) on a layout the fused kernel does not serve (three unequal
    layers: five launches, run-time dispatch in the GEMM epilogues) and as the weight-initialisation rule of recurrent layers."""
    kw = dict(dimS=9, dimA=3, bounded=[0, 1, 0], hidden=(24, 16, 8), nnFunc=func, batchSize=12, maxTotObsNum=1500, randSeed=5)
    G, O = _pair(hip_api, kw, synth_cfg(seed=3, dimS=9, dimA=3, lenMin=3, lenMax=30, pTerm=0.3), 30)
    for _ in range(4):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(17); O.step(17)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    kw = dict(dimS=6, dimA=2, hidden=(16, 16), nnFunc=func, batchSize=8, maxTotObsNum=1500, randSeed=5, nn_type=capi.NN_LSTM, nnBPTTseq=4)
    G, O = _pair(hip_api, kw, synth_cfg(seed=3, dimS=6, dimA=2, lenMin=3, lenMax=30, pTerm=0.3), 30)
    assert np.array_equal(G.get_params()[0], O.get_params()[0])          # Layer::initialize with this function's fan-in / fan-out rule
    G.step(3); O.step(3)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(28))
def test_random_configurations_match_oracle(hip_api, seed):
    """Randomly drawn configurations (dense / LSTM / MGU layers x the three heads x odd sizes): the library against the
    oracle over a few single steps and a replayed stretch."""
    kw, sc = _random_config(np.random.default_rng(1000 + seed))
    n_eps = 40
    try:
        G, O = _pair(hip_api, kw, synth_cfg(**sc), n_eps)
    except capi.HlError as e:                # too few transitions for the batch: both sides must say so
        assert e.status == 5, (kw, sc, str(e))
        return
    for _ in range(3):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(12); O.step(12)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT)), (kw, sc)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32, (kw, sc)
    assert abs(G.scalars().beta - O.scalars().beta) <= 1e-9 * O.scalars().beta


REC_SHAPES = [  # (layer type, hidden, dimS, bptt, batch): every instantiation / fall-back of the recurrent kernels
    ("lstm", (13,), 3, 5, 9),               # one layer, cells not a multiple of 4 (padded chunks, 4 lanes per gate)
    ("lstm", (5, 5, 5), 1, 3, 17),          # three layers, 8 lanes per gate
    ("lstm", (24, 16, 8, 8), 7, 6, 12),     # four layers: the general (runtime layer count) bodies
    ("lstm", (32, 32), 4, 16, 33),          # the RACER_RNN.json network: cells known at compile time
    ("lstm", (32, 32), 4, 16, 128, 1),      # ... and its exact row: batch 128, ONE action (partially observable cart-pole)
    ("mgu", (32, 32), 4, 16, 128, 1),       # the same MDP with nnType left at its default
    ("lstm", (64, 64), 30, 4, 8),           # one lane per gate
    ("lstm", (64, 64), 200, 3, 6),          # weights too large for LDS: the one-thread-per-gate kernels
    ("lstm", (48, 40), 130, 11, 5),         # BPTT window + weights beyond the LDS budget of the BPTT kernel only
    ("lstm", (32, 32), 4, 16, 64),          # 64 x 17 = 1088 rows: the weight gradients are split over the rows (splitk_reduce_kernel)
    ("mgu", (24, 16), 6, 15, 70),           # the same for the three weight-gradient problems of an MGU layer
    ("mgu", (24, 16, 8, 8), 7, 6, 12),
    ("mgu", (32, 32), 4, 16, 33),           # two layers of 32 cells: one wavefront per (sample, layer), as for the LSTM
    ("mgu", (32, 32), 13, 16, 64),          # ... inputs padded to 16, split weight gradients
    ("mgu", (32, 32), 32, 5, 20),
    ("mgu", (13,), 3, 5, 9),
    ("mgu", (64, 64), 200, 3, 6),
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", REC_SHAPES, ids=lambda sh: "%s-%s-dS%d" % (sh[0], "x".join(map(str, sh[1])), sh[2]))
def test_recurrent_kernel_variants_match_oracle(hip_api, shape):
    """The recurrent kernels are instantiated per layer count / cell count and fall back to slower bodies when weights or the
    window's activations do not fit in LDS: one configuration per variant, against the oracle."""
    kind, hidden, dS, bptt, batch = shape[:5]
    dA = shape[5] if len(shape) > 5 else 2
    kw = dict(dimS=dS, dimA=dA, bounded=[1, 0][:dA], hidden=hidden, nnFunc="Tanh", batchSize=batch, maxTotObsNum=8000, randSeed=5,
              nn_type=capi.NN_LSTM if kind == "lstm" else capi.NN_MGU, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=bptt)
    if len(shape) > 5:      # settings/RACER_RNN.json
        kw.update(gamma=0.99, nnLambda=1e-6, explNoise=0.1, epsAnneal=0.0)
    G, O = _pair(hip_api, kw, synth_cfg(seed=21, dimS=dS, dimA=dA, lenMin=2, lenMax=30, pTerm=0.5), 60 if batch < 100 else 120)
    for _ in range(3):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(10); O.step(10)                      # replayed: the sampler of the next step rides along the head kernel
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_random_fused_kernel_shapes_match_oracle(hip_api, seed):
    """Shapes served by the fused forward / head / dX kernel, drawn at random: width 16..256, 1..32 state components, 1..7
    actions with mixed bounds, batches 1..300 (partial panels, many next-state rows with short truncated episodes)."""
    rng = np.random.default_rng(3000 + seed)
    H = int(rng.choice([16, 32, 64, 128, 256]))
    dS, dA = int(rng.integers(1, 33)), int(rng.integers(1, 8))
    kw = dict(dimS=dS, dimA=dA, bounded=[int(b) for b in rng.integers(0, 2, dA)], hidden=(H, H),
              nnFunc=["SoftSign", "Tanh", "Relu"][int(rng.integers(3))], batchSize=int(rng.integers(1, 300)),
              maxTotObsNum=20000, randSeed=int(rng.integers(1, 1000)), clipImpWeight=float(rng.choice([0.7, 2.0, 4.0])))
    sc = dict(seed=int(rng.integers(1, 1000)), dimS=dS, dimA=dA, lenMin=2, lenMax=int(rng.integers(3, 60)),
              pTerm=float(rng.choice([0.0, 0.5, 1.0])))
    try:
        G, O = _pair(hip_api, kw, synth_cfg(**sc), 120)
    except capi.HlError as e:
        assert e.status == 5, (kw, sc, str(e))
        return
    for _ in range(2):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(9); O.step(9)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT)), (kw, sc)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32, (kw, sc)


@pytest.mark.gpu
def test_error_paths_fail_loudly(hip_api):
    """Call-sequence and size errors come back as status codes, never as silent work
    (reference: die() in Learner_approximator.cpp:38-41 for a too small replay)."""
    cfg = capi.make_config(dimS=4, dimA=2, bounded=[1, 1], hidden=(16, 16), batchSize=32, maxTotObsNum=200, randSeed=1)
    L = hip_learner(hip_api, cfg)
    L.init_weights()
    with pytest.raises(capi.HlError) as e:
        L.initialize()                             # empty replay
    assert e.value.status == 5                     # HL_ERR_TOO_FEW_DATA
    sc = synth_cfg(seed=5, dimS=4, dimA=2, lenMin=3, lenMax=3, pTerm=0.0)
    fill_synth(L, sc, 4)                           # 8 transitions < batch 32
    with pytest.raises(capi.HlError) as e:
        L.step(1)                                  # step before initialize
    assert e.value.status == 4                     # HL_ERR_STATE
    L.initialize()
    with pytest.raises(capi.HlError) as e:
        L.step(1)
    assert e.value.status == 5 and "minTotObsNum" in str(e.value)
    with pytest.raises(capi.HlError) as e:
        L.step_end()                               # without step_begin
    assert e.value.status == 4


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,sc_kw", [
    (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=64, maxTotObsNum=20000, randSeed=1),
     dict(seed=9, dimS=17, dimA=6, lenMin=50, lenMax=90, pTerm=0.2)),
    (dict(dimS=9, dimA=3, bounded=[0, 0, 0], hidden=(24, 16, 8), nnFunc="Tanh", batchSize=8, maxTotObsNum=1000, randSeed=5),
     dict(seed=3, dimS=9, dimA=3, lenMin=3, lenMax=30, pTerm=0.3)),
])
def test_rollout_forward_matches_oracle(hip_api, cfg_kw, sc_kw):
    """hl_forward (what RACER::selectAction reads, RACER.cpp:30-47): network outputs for raw states
    with the current weights and state scaling, before and after training steps; more states than
    minibatch rows (chunked) and a single state."""
    sc = synth_cfg(**sc_kw)
    G, O = _pair(hip_api, cfg_kw, sc, 40)
    rng = np.random.default_rng(5)
    for n in (1, 7, 3 * cfg_kw["batchSize"] + 5):
        st = rng.standard_normal((n, cfg_kw["dimS"])).astype(np.float32) * 2 + 0.3
        og, oo = G.forward(st), O.forward(st)
        assert og.shape == (n, G.nOut) and relinf(og, oo) < TOL32
    G.step(20); O.step(20)
    st = rng.standard_normal((33, cfg_kw["dimS"])).astype(np.float32)
    assert relinf(G.forward(st), O.forward(st)) < TOL32
    G.step(3); O.step(3)            # the forward pass in between leaves the training path untouched
    _compare_step(G, O)


# ---------------------------------------------------------------------------------------------
# BASELINE.json size: 1M-transition replay, 17/6, 2x256, B=256
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden", [(capi.NN_LSTM, (32, 32)), (capi.NN_MGU, (32, 32)),      # one wavefront per layer (weights in registers)
                                         (capi.NN_LSTM, (24, 16)), (capi.NN_MGU, (24, 16, 8))],   # the general kernels
                         ids=["lstm-2x32", "mgu-2x32", "lstm-24x16", "mgu-24x16x8"])
def test_recurrent_acting_matches_oracle(hip_api, kind, hidden):
    """hl_forward_sequence: the agent's last min(nnBPTTseq, t) + 1 states forwarded from a zero recurrent state
    (MemoryBuffer::agentToMinibatch + Approximator::forward(agent)), after some training, windows of every length."""
    cfg_kw = dict(dimS=6, dimA=2, bounded=[1, 0], hidden=hidden, nnFunc="Tanh", batchSize=16, maxTotObsNum=5000, randSeed=51,
                  adv_kind=capi.ADV_GAUSSIAN, nn_type=kind, nnBPTTseq=8)
    G, O = _pair(hip_api, cfg_kw, synth_cfg(seed=43, dimS=6, dimA=2, lenMin=5, lenMax=40, pTerm=0.5), 40)
    G.step(5); O.step(5)
    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 9):
        S = rng.normal(size=(n, 6)).astype(np.float32)
        og, oo = G.forward_sequence(S), O.forward_sequence(S)
        assert relinf(og, oo) < TOL32, (n, og, oo)
    with pytest.raises(capi.HlError):
        G.forward_sequence(rng.normal(size=(10, 6)).astype(np.float32))      # more than nnBPTTseq + 1 steps
    with pytest.raises(capi.HlError):
        G.forward(rng.normal(size=(1, 6)).astype(np.float32))                # stateless forward of a recurrent net


CONV_ATARI = [(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]     # apps/OpenAI_gym_atari/exec.py:114-117


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["conv_small.bin", "racer_atari.bin", "nature_dqn.bin"])
def test_conv_steps_follow_reference_fixture(hip_api, name):
    """BASELINE config 5 (RACER_atari.json: 84x84 frames x (1 + 3 appended), four SoftSign convolutions, dense 512 + parametric
    residual, discrete RACER head, batch 128) and a two-layer variant: the (episode, t >= nAppendedObs) pairs of the compiled
    reference's harness through the device path -- stacked gather, implicit-GEMM convolutions forward / dX / dW on MFMA,
    Adam -- against the reference's own taps."""
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc="Tanh"))
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    setup_from_fixture(L, fx)
    w0 = L.get_params()[0]
    assert fx_vec_dev(fx, "W0", w0) < 1e-12 and np.array_equal(w0[::53], fx["W0_sub"])      # Conv2DLayer::initialize draw order
    m, sc, r = L.get_scaling()
    assert np.allclose(np.concatenate([m, sc, r]), fx["scaling0"], rtol=2e-7, atol=1e-7)
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        flat = flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        order = np.argsort(flat, kind="stable")
        L.step(1, flat=flat[order])
        assert np.array_equal(L.readback(capi.TAP_TAG), fx[sk + "tag"][order]) and np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"][order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"][order]) < TOL32
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"][order])
        if sk + "gradSum" in fx or sk + "gradSum_sub" in fx:
            assert fx_vec_dev(fx, sk + "gradSum", L.readback(capi.TAP_GRADSUM)) < TOL32
        if sk + "W_sub" in fx:
            w, m1, m2 = L.get_params()
            assert fx_vec_dev(fx, sk + "W", w) < TOL32
            assert fx_vec_dev(fx, sk + "M1", m1) < 2 * TOL32 and fx_vec_dev(fx, sk + "M2", m2) < 2 * TOL32
        assert L.scalars().nFarPolicySteps == fx["traj_nfar"][k - 1]
    assert fx_vec_dev(fx, "Wfinal", L.get_params()[0]) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps,steps", [
    # two convolutions behind 1 + 3 stacked observations; short episodes: steps t < 3 (first frame repeated) and
    # truncated next states (rows >= B) are sampled
    (dict(dimS=256, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=5, nAppendedObs=3, conv=[(8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)],
          hidden=(48,), nnFunc="Tanh", batchSize=24, maxTotObsNum=2000, randSeed=3),
     dict(seed=5, dimS=256, dimA=1, lenMin=3, lenMax=9, pTerm=0.3), 60, 6),
    # stride-2 layer with 8 -> 16 channels (the second RACER_atari layer), continuous V-RACER head, two dense layers
    (dict(dimS=800, dimA=2, bounded=[1, 0], nAppendedObs=3, conv=[(20, 20, 8, 16, 6, 2)], hidden=(40, 24), batchSize=16,
          maxTotObsNum=1500, randSeed=4),
     dict(seed=6, dimS=800, dimA=2, lenMin=4, lenMax=20, pTerm=0.5), 30, 5),
    # a strided layer BEHIND another one: its input gradient runs per parity class of the input position (filter and image sizes
    # multiples of the stride) ...
    (dict(dimS=576, dimA=2, nAppendedObs=0, conv=[(12, 12, 4, 8, 3, 1), (10, 10, 8, 16, 4, 2)], hidden=(32,), nnFunc="Tanh", batchSize=12,
          maxTotObsNum=900, randSeed=2),
     dict(seed=4, dimS=576, dimA=2, lenMin=3, lenMax=12, pTerm=0.4), 25, 5),
    # ... or, when they are not (11 x 11 image, filter 3, stride 2), through the all-taps kernel
    (dict(dimS=507, dimA=2, nAppendedObs=0, conv=[(13, 13, 3, 8, 3, 1), (11, 11, 8, 12, 3, 2)], hidden=(32,), nnFunc="Tanh", batchSize=12,
          maxTotObsNum=900, randSeed=2),
     dict(seed=4, dimS=507, dimA=2, lenMin=3, lenMax=12, pTerm=0.4), 25, 5),
    # odd geometry: 3 input channels, 5 filters, 7x9 image, stride 1, filter 3 (channel and position tiles partly empty)
    (dict(dimS=189, dimA=2, nAppendedObs=0, conv=[(9, 7, 3, 5, 3, 1)], hidden=(32,), nnFunc="SoftSign", batchSize=10,
          maxTotObsNum=800, randSeed=6),
     dict(seed=8, dimS=189, dimA=2, lenMin=3, lenMax=12, pTerm=0.4), 25, 5),
    # appended observations without convolutions: dense layers on the stacked state
    (dict(dimS=6, dimA=2, nAppendedObs=2, hidden=(32, 32), batchSize=16, maxTotObsNum=900, randSeed=9),
     dict(seed=2, dimS=6, dimA=2, lenMin=2, lenMax=15, pTerm=0.5), 40, 8),
])
def test_conv_and_appended_observations_match_oracle(hip_api, cfg_kw, sc_kw, n_eps, steps):
    sc = synth_cfg(**sc_kw)
    G, O = _pair(hip_api, cfg_kw, sc, n_eps)
    for _ in range(steps):
        G.step(1); O.step(1)
        _compare_step(G, O)
    assert (G.readback(capi.TAP_TSTEP) >= 0).all()
    G.step(21); O.step(21)                      # replayed graphs (16 + 4 + 1), riders draw the next minibatch
    _compare_step(G, O)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    st = np.random.default_rng(0).normal(size=(3, G.dIn)).astype(np.float32)      # rollout inference on stacked raw states
    assert relinf(G.forward(st), O.forward(st)) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps", [
    (dict(dimS=17, dimA=6, hidden=(64, 64), batchSize=32, maxTotObsNum=3000, randSeed=8),          # fused path
     dict(seed=21, dimS=17, dimA=6, lenMin=5, lenMax=30, pTerm=0.3), 60),
    (dict(dimS=9, dimA=3, bounded=[0, 1, 0], hidden=(24, 16), nnFunc="Tanh", batchSize=8, maxTotObsNum=1500, randSeed=5),   # five launches
     dict(seed=3, dimS=9, dimA=3, lenMin=3, lenMax=30, pTerm=0.3), 40),
])
def test_chained_replays_carry_the_next_minibatch_across_calls(hip_api, cfg_kw, sc_kw, n_eps):
    """Every replayed step draws the minibatch of the step after it, also the last one of a call: the next call starts from
    it (odd and even call lengths alternate the buffer), and whatever invalidates it -- a generator read-out, new episodes,
    explicit indices, a rollout forward -- puts the generator back first.  Same trajectory as the oracle throughout."""
    sc = synth_cfg(**sc_kw)
    G, O = _pair(hip_api, cfg_kw, sc, n_eps)
    nxt = n_eps
    for i, n in enumerate([1, 1, 1, 2, 5, 20, 3, 1, 7, 16, 1, 33]):
        G.step(n); O.step(n)
        if i % 4 == 3:       # generator read-out: the minibatch drawn ahead is discarded, the state is the oracle's
            assert np.array_equal(G.get_rng_state(), O.get_rng_state()), i
        if i == 5:           # new episodes between two calls
            for L in (G, O):
                fill_synth(L, sc, 3, first=nxt)
            nxt += 3
        if i == 8:           # rollout inference borrows a minibatch buffer
            st = np.random.default_rng(0).normal(size=(4, cfg_kw["dimS"])).astype(np.float32)
            assert relinf(G.forward(st), O.forward(st)) < TOL32
        if i == 9:           # explicit indices
            flat = np.sort(np.random.default_rng(1).choice(int(G.scalars().nStoredSteps), G.B, replace=False)).astype(np.int64)
            G.step(1, flat=flat); O.step(1, flat=flat)
    _compare_step(G, O)
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < TOL32
    assert G.scalars().nGradSteps == O.scalars().nGradSteps
    assert abs(G.scalars().beta - O.scalars().beta) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, dict(hidden=(24, 16, 8), nnFunc="Tanh")], ids=["fused-2x32", "generic-24x16x8"])
def test_announced_call_sizes_replay_as_one_graph_with_a_completion_stamp(hip_api, extra):
    """hl_prepare_steps(n): a call of n steps is ONE graph whose last node stamps a pinned host word that hl_sync polls.
    Bit-identical to the same calls served from the stock graph sizes; the stamp path survives new episodes, a 1000th-step
    sweep inside a call (served piecewise), interleaved other calls, and call sizes that prepare themselves after three calls."""
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=3000, randSeed=42)
    cfg_kw.update(extra)
    sc = synth_cfg(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5)
    A, O = _pair(hip_api, cfg_kw, sc, 60)
    B, _ = _pair(hip_api, cfg_kw, sc, 60)
    A.prepare_steps(20); A.prepare_steps(7)
    nxt = 60
    for i, n in enumerate([5, 20, 20, 7, 20, 3, 20, 11, 11, 11, 11, 11, 20]):
        A.step(n); B.step(n); O.step(n)
        A.sync(); B.sync()
        if i % 3 == 1:
            assert np.array_equal(A.get_params()[0], B.get_params()[0]), i
        if i == 4:
            for L in (A, B, O):
                fill_synth(L, sc, 2, first=nxt)
            nxt += 2
    for _ in range(42):      # across the 1000th step
        A.step(20); B.step(20); O.step(20)
    A.sync(); B.sync()
    assert A.scalars().nGradSteps == B.scalars().nGradSteps == O.scalars().nGradSteps
    for a, b in zip(A.get_params(), B.get_params()):
        assert np.array_equal(a, b)
    assert np.array_equal(A.get_rng_state(), B.get_rng_state())
    assert np.array_equal(A.get_rng_state(), O.get_rng_state())
    assert A.scalars().beta == B.scalars().beta
    assert relinf(A.get_params()[0], O.get_params()[0]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, dict(hidden=(24, 16, 8), nnFunc="Tanh"), dict(hidden=(16, 16), nnFunc="Tanh", nn_type=capi.NN_MGU, nnBPTTseq=4)],
                         ids=["fused-2x32", "generic-24x16x8", "mgu-2x16"])
def test_replicas_speak_one_wire_protocol_on_every_path(hip_api, extra):
    """With a communicator attached every step -- replayed or eager, with or without an eviction, with explicit indices --
    issues exactly ONE all-reduce (gradient || counters), the 1000th step one more (moments): replicas that take different
    paths (their replays fill differently) still pair their collectives one to one.  (Networks off the fused path replay as
    graphs with the captured collective too.)"""
    import ctypes as C
    cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=600, randSeed=42)
    cfg_kw.update(extra)
    sc = synth_cfg(seed=7, dimS=5, dimA=2, lenMin=5, lenMax=40, pTerm=0.5)
    A, _ = _pair(hip_api, cfg_kw, sc, 30)
    Bq = hip_learner(hip_api, capi.make_config(**cfg_kw))
    Bq.init_weights(); fill_synth(Bq, sc, 30)
    raw = (C.c_uint8 * 128)()
    assert hip_api.fn("comm_unique_id")(raw) == 0
    Bq.comm_init(bytes(raw))
    Bq.initialize()
    coll = hip_api.lib.hl_debug_collectives
    coll.restype = C.c_int64; coll.argtypes = [C.c_void_p]
    c0 = coll(Bq.h)
    done = 0
    nxt = 30
    for n in (1, 20, 3, 64, 7):
        A.step(n); Bq.step(n); done += n
        assert coll(Bq.h) - c0 == done, n
        for L in (A, Bq):                      # appended episodes push the replay over its budget: eager steps with evictions
            fill_synth(L, sc, 4, first=nxt)
        nxt += 4
        assert np.array_equal(A.get_params()[0], Bq.get_params()[0]), n
    flat = np.sort(np.random.default_rng(1).choice(int(A.scalars().nStoredSteps), A.B, replace=False)).astype(np.int64)
    A.step(1, flat=flat); Bq.step(1, flat=flat); done += 1
    assert coll(Bq.h) - c0 == done
    rest = 1000 - done
    A.step(rest); Bq.step(rest)                # ends on the 1000th step: + the moments all-reduce
    assert coll(Bq.h) - c0 == 1000 + 1
    assert np.array_equal(A.get_params()[0], Bq.get_params()[0])
    assert A.scalars().beta == Bq.scalars().beta and A.scalars().nFarPolicySteps == Bq.scalars().nFarPolicySteps
    assert np.array_equal(A.get_rng_state(), Bq.get_rng_state())


@pytest.mark.gpu
def test_episodes_and_actions_from_other_threads_while_training(hip_api):
    """The learner's lock (the reference's dataset_mutex, MemoryBuffer.h:55): env-service threads hand over finished episodes and
    ask for network outputs while the training thread steps.  Nothing is lost, the counters add up, and -- the interleaving
    being whatever it was -- a second learner fed the same episodes in the order the first one stored them, then stepped
    alone, holds the same replay contents."""
    import threading
    cfg_kw = dict(dimS=17, dimA=6, hidden=(64, 64), batchSize=32, maxTotObsNum=50000, randSeed=8)
    sc = synth_cfg(seed=21, dimS=17, dimA=6, lenMin=5, lenMax=30, pTerm=0.3)
    G = hip_learner(hip_api, capi.make_config(**cfg_kw))
    G.init_weights(); fill_synth(G, sc, 60); G.initialize()
    errs = []

    def feeder(first, n):
        try:
            for e in range(first, first + n):
                G.append_episode(**synth_episode(sc, e))
                G.forward(np.zeros((2, 17), np.float32))
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)

    ths = [threading.Thread(target=feeder, args=(1000 * (i + 1), 80)) for i in range(3)]
    for t in ths:
        t.start()
    steps = 0
    while any(t.is_alive() for t in ths):
        G.step(3); steps += 3
    for t in ths:
        t.join()
    G.step(5); steps += 5
    assert not errs, errs
    nS, nE, nG, seenS, seenE = G.counts()
    assert nE == 60 + 240 and nG == steps and seenE == 300
    lens = [G.episode_info(p)[1] for p in range(nE)]
    assert nS == sum(lens) - nE and seenS == nS
    sca = G.scalars()
    assert sca.nStoredSteps == nS and sca.nGradSteps == steps
    tags = sorted(G.episode_info(p)[0] for p in range(nE))
    assert tags == sorted(list(range(60)) + [1000 * (i + 1) + k for i in range(3) for k in range(80)])
    # every stored episode holds its own data (nothing interleaved inside the staging buffers)
    for p in (0, 7, 150, nE - 1):
        tag, N, term = G.episode_info(p)
        ref = synth_episode(sc, tag)
        pk = G.pack_episode(p).reshape(-1)
        assert np.array_equal(pk[:17], ref["states"][0]) and N == ref["rewards"].size


@pytest.mark.gpu
def test_importance_weight_histogram(hip_api):
    """hl_impweight_histogram (MemoryProcessing::histogramImportanceWeights, MemoryProcessing.cpp:353-389) on the fixture state
    of the compiled reference, and against the oracle on a larger replay."""
    fx = load_fixture("hist_small.bin")
    L = hip_learner(hip_api, fixture_config(fx))
    setup_from_fixture(L, fx)
    for k in range(1, 3):      # the reference's samples of the tapped steps, its own afterwards
        flat = flat_for(L, fx["s%d_tag" % k], fx["s%d_t" % k])
        L.step(1, flat=np.sort(flat))
    text, cnt = L.impweight_histogram()
    assert cnt.sum() == L.scalars().nStoredSteps
    assert text.splitlines()[:4] == bytes(bytearray(fx["impw_histogram"])).decode().splitlines()[:4]     # header + bin centres
    cfg_kw = dict(dimS=17, dimA=6, hidden=(64, 64), batchSize=64, maxTotObsNum=20000, randSeed=8)
    G, O = _pair(hip_api, cfg_kw, synth_cfg(seed=21, dimS=17, dimA=6, lenMin=5, lenMax=60, pTerm=0.3, muSpread=0.8), 300)
    G.step(200); O.step(200)
    tg, cg = G.impweight_histogram(); to, co = O.impweight_histogram()
    assert cg.sum() == co.sum() == G.scalars().nStoredSteps
    assert np.abs(cg - co).sum() <= 4 and (cg[1:80] > 0).sum() > 20      # (a weight within 1e-6 of a bin edge may change bins)
    assert tg.splitlines()[3] == to.splitlines()[3]


@pytest.fixture(scope="module")
def full_size(hip_api):
    cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=1000000, randSeed=42)
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=201, lenMax=201, pTerm=0.0)
    G, O = _pair(hip_api, cfg_kw, sc, 5000)
    return G, O


def test_full_size_sampler_properties(full_size):
    G, O = full_size
    assert G.scalars().nStoredSteps == 1000000
    seen = []
    for _ in range(20):
        G.step(1); O.step(1)
        flat = G.readback(capi.TAP_FLAT)
        assert np.array_equal(flat, O.readback(capi.TAP_FLAT))
        assert flat.min() >= 0 and flat.max() < 1000000
        assert np.all(np.diff(flat) > 0)                       # sorted, unique
        t = G.readback(capi.TAP_TSTEP); pos = G.readback(capi.TAP_EPISODE)
        assert np.array_equal(pos * 200 + t, flat)             # IDtoSeqStep on equal-length episodes
        seen.append(flat)
    allf = np.concatenate(seen)
    assert 0.45e6 < allf.mean() < 0.55e6                       # uniform over the buffer
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())


def test_full_size_steps_match_oracle(full_size):
    """B=256 steps on the 1M-transition buffer: gradients within 1e-5, masks / indices exact."""
    G, O = full_size
    for _ in range(3):
        G.step(1); O.step(1)
        _compare_step(G, O)
    assert relinf(G.get_params()[0], O.get_params()[0]) < TOL32
    mG, sG, rG = G.get_scaling(); mO, sO, rO = O.get_scaling()
    assert np.allclose(np.concatenate([mG, sG, rG]), np.concatenate([mO, sO, rO]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("algo", ["PERerr", "PERrank"])
def test_full_size_prioritised_sampler(hip_api, algo):
    """The prioritised samplers on the 1M-transition replay: the sequential normalisation / cumulative table runs over a million
    values (977 blocks of the one-wavefront chain), the ranking over a million keys -- minibatches, generator and the errors the
    steps write back equal the oracle's; sample indices sorted, unique and inside the buffer."""
    cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=1000000, randSeed=42, dataSamplingAlgo=algo)
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=201, lenMax=201, pTerm=0.0)
    G, O = _pair(hip_api, cfg_kw, sc, 5000)
    for k in range(4):
        G.step(1); O.step(1)
        flat = G.readback(capi.TAP_FLAT)
        assert np.array_equal(flat, O.readback(capi.TAP_FLAT)), k
        assert flat.min() >= 0 and flat.max() < 1000000 and np.all(np.diff(flat) > 0)
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    _compare_step(G, O)


def test_full_size_retrace_round_trip(full_size):
    """Property at full size: Retrace of a truncated episode satisfies its own recursion
    Q_t = r_{t+1} + gamma (V_{t+1} + min(1, rho_{t+1}) (Q_{t+1} - V_{t+1})) on the values the GPU holds."""
    G, O = full_size
    _, _, r3 = G.get_scaling()
    for pos in (0, 1234, 4999):
        Q = G.episode_field(pos, capi.EP_RETURN); V = G.episode_field(pos, capi.EP_VALUE)
        W = G.episode_field(pos, capi.EP_IMPW)
        tag, N, term = G.episode_info(pos)
        R = synth_episode(synth_cfg(seed=7, dimS=17, dimA=6, lenMin=201, lenMax=201, pTerm=0.0), tag)["rewards"]
        assert np.allclose(Q, O.episode_field(pos, capi.EP_RETURN), rtol=1e-4, atol=1e-4)
        # the sweep ran at initialize(); steps sampled since then changed V / rho of a few entries (their stored Q_t stays
        # as it was until the next sweep), so the recursion is asserted exactly where nothing was touched: rho_{t+1} still
        # at its insert-time value 1 -- for the last transition, whose successor is the truncated end state (rho = 0,
        # Q = V), the sampled step itself untouched
        rs = ((R - np.float64(r3[0])) * np.float64(r3[1])).astype(np.float32)
        g = np.float32(0.995)
        checked = 0
        for t in range(N - 2, -1, -1):
            untouched = (W[t + 1] == 1.0) if t + 1 < N - 1 else (W[t] == 1.0 and W[t + 1] == 0.0)
            if not untouched:
                continue
            w = min(np.float32(1), W[t + 1])
            rhs = rs[t + 1] + g * (V[t + 1] + w * (Q[t + 1] - V[t + 1]))
            assert abs(Q[t] - rhs) <= 2e-5 * max(1.0, abs(rhs)), (pos, t, Q[t], rhs)
            checked += 1
        assert checked > N // 2
        assert np.isfinite(Q).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_far_policy_count_over_more_episodes_than_the_register_walk_holds(hip_api, fused):
    """7000 short episodes: 28 table positions per thread of the count's walk, more than its 24 registers (and than the LDS copy of
    the bookkeeping rider) -- the general walk over memory, inside single steps (bookkeeping pass) and inside replayed calls (the
    rider of the next step's kernel on the fused path, of the dW launch on the generic one).  Count, beta and the generator state
    equal the oracle's; a low clip keeps many steps far from the behaviour policy."""
    cfg_kw = dict(dimS=3, dimA=1, bounded=[1], hidden=(16, 16) if fused else (24, 16), batchSize=64, maxTotObsNum=60000, randSeed=77,
                  clipImpWeight=0.3, learnrate=1e-3)
    G, O = _pair(hip_api, cfg_kw, synth_cfg(seed=21, dimS=3, dimA=1, lenMin=3, lenMax=6, pTerm=0.3, muSpread=0.5), 7000)
    assert G.scalars().nStoredEps == 7000
    for k in range(4):
        G.step(1); O.step(1)
        _compare_step(G, O)
        assert G.scalars().nFarPolicySteps == O.scalars().nFarPolicySteps and G.scalars().beta == pytest.approx(O.scalars().beta, rel=1e-12)
    G.step(24); O.step(24)          # 16 + 8 replayed steps: the deferred count between them
    sg, so = G.scalars(), O.scalars()
    assert sg.nFarPolicySteps == so.nFarPolicySteps and sg.nFarPolicySteps > 0
    assert abs(sg.beta - so.beta) <= 1e-12 * so.beta
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())


@pytest.mark.gpu
def test_several_learners_of_one_process_step_side_by_side(hip_api):
    """Four independent single-replica learners (one thread each, as a multi-agent run holds them) replay 20-step calls at the same
    time on the one device: the kernels of different learners interleave, the in-kernel waits of each (panel barrier, beta handed
    from the rider of the next fused kernel) only ever concern workgroups of their own launch -- no time-out, and every learner ends
    bit-identical to the same learner stepped alone."""
    import threading
    cfg_kw = dict(dimS=17, dimA=6, hidden=(64, 64), batchSize=64, maxTotObsNum=50000)
    sc = synth_cfg(seed=31, dimS=17, dimA=6, lenMin=20, lenMax=60, pTerm=0.3)

    def make(seed):
        L = hip_learner(hip_api, capi.make_config(randSeed=seed, **cfg_kw))
        L.init_weights(); fill_synth(L, sc, 120); L.initialize(); L.prepare_steps(20)
        return L

    def state(L):
        w, m1, m2 = L.get_params(); s = L.scalars()
        return w.tobytes(), m1.tobytes(), s.beta, s.nFarPolicySteps, L.get_rng_state().tobytes()

    alone = []
    for seed in range(4):
        L = make(100 + seed)
        for _ in range(40):
            L.step(20)
        alone.append(state(L)); L.close()
    Ls = [make(100 + seed) for seed in range(4)]
    errs = []

    def run(L):
        try:
            for _ in range(40):
                L.step(20)
            L.sync()
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)

    ths = [threading.Thread(target=run, args=(L,)) for L in Ls]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for L, ref in zip(Ls, alone):
        assert state(L) == ref
