"""CPU suite: pins the oracle (oracle/port, CPU restatement) against golden fixtures that were
produced by the compiled reference itself (tests/golden/make_golden.sh).  No GPU needed."""
import numpy as np
import pytest

from oracle_api import oracle_learner, synth_episode
from parity import (load_fixture, fixture_config, fixture_synth, fixture_arrival, setup_from_fixture, relinf,
                    episode_arrays_by_tag, fixture_arrays_by_tag, stats_line, lines_agree, fx_vec_dev, flat_for)
from smarties_amd import capi

ACT_FIXTURES = ["act_%s.bin" % f for f in ("LRelu", "Sigm", "HardSign", "SoftPlus", "ExpPlus", "Exp")]     # the other names of makeFunction (Functions.h:643-668)
PER_FIXTURES = ["sample_%s.bin" % f for f in ("PERrank", "PERerr", "PERseq")]      # dataSamplingAlgo (Sampling.cpp:101-296)
EVICT_FIXTURES = ["evict_%s.bin" % f for f in ("farpolfrac", "maxkldiv", "minerror")]      # ERoldSeqFilter (MemoryProcessing.cpp:261-298)
FUNC_OF = {"pomdp_encoder.bin": "Tanh", "lstm_wide.bin": "Tanh", "mgu_wide.bin": "Tanh", "hp_odd.bin": "Tanh", "discrete_lstm.bin": "Tanh", "gauss_mgu.bin": "Tanh", "one_layer_relu.bin": "Relu", "deep_tanh.bin": "Tanh", "racer_lstm.bin": "Tanh", "vracer_mgu.bin": "Tanh", **{n: n[4:-4] for n in ACT_FIXTURES}}


def make(name):
    fx = load_fixture(name)
    cfg = fixture_config(fx, nnFunc=FUNC_OF.get(name), episode_order=capi.ORDER_REFERENCE)
    L = oracle_learner(cfg)
    return fx, L


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "ns_shape.bin", "racer_gauss.bin", "racer_discrete.bin", "racer_lstm.bin", "vracer_mgu.bin", "threads3.bin"] + ACT_FIXTURES)
def test_layout_init_and_rng_match_reference(name):
    """Parameters blob layout (Parameters.h:159-176), Layer::initialize draw order and the
    libstdc++ uniform_real_distribution<float> restatement: weights and RNG state bit-exact."""
    fx, L = make(name)
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    assert np.array_equal(lay["nW"], fx["nWeights"]) and np.array_equal(lay["nB"], fx["nBiases"])
    L.init_weights()
    w, m1, m2 = L.get_params()
    assert np.array_equal(w, fx["W0"])
    assert not m1.any() and not m2.any()
    assert np.array_equal(L.get_rng_state(), fx["rng_before_init"])


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "ns_shape.bin", "racer_gauss.bin", "racer_discrete.bin", "racer_lstm.bin", "vracer_mgu.bin"])
def test_initialize_matches_reference(name):
    """Learner::initializeLearner: beta after the init updateCounters, exact reward/state
    statistics, Retrace of every episode after rescaling."""
    fx, L = make(name)
    setup_from_fixture(L, fx)
    s = L.scalars()
    assert s.nStoredSteps == int(fx["cfg"][7])
    assert s.beta == fx["beta0"][0] and s.CmaxRet == fx["cmax0"][0]
    m, sc, r = L.get_scaling()
    assert np.array_equal(np.concatenate([m, sc, r]), fx["scaling0"])
    lens = {e: synth_episode(fixture_synth(fx), e)["rewards"].size for e in range(int(fx["cfg"][3]))}
    mine = episode_arrays_by_tag(L, capi.EP_RETURN)
    ref = fixture_arrays_by_tag(fx, "ret0_tags", "ret0", lens)
    for tag, arr in ref.items():
        assert np.allclose(mine[tag], arr, rtol=1e-6, atol=1e-6), tag
    assert np.array_equal(L.get_rng_state(), fx["rng0"])


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "ns_shape.bin", "racer_gauss.bin", "racer_discrete.bin", "racer_lstm.bin", "vracer_mgu.bin", "threads3.bin", "hp_odd.bin", "hp_lowclip.bin", "discrete_lstm.bin", "one_layer_relu.bin", "gauss_mgu.bin", "crowded_sampler.bin", "moving_replay.bin", "lstm_wide.bin", "mgu_wide.bin", "pomdp_encoder.bin"] + ACT_FIXTURES + EVICT_FIXTURES + PER_FIXTURES)
def test_steps_match_reference(name):
    """Every tapped step: sampled flat indices / (episode, t) bit-exact (mt19937 + Lemire
    uniform_int + sort/unique/redraw + the reference's std::sort episode permutation); network
    outputs, rho, D_KL, delta-Q, output gradients to 1e-6 relative (f64 head: 1e-12); ReF-ER mask
    exact; summed weight gradient, weights and Adam moments to 1e-5 of the infinity norm."""
    fx, L = make(name)
    setup_from_fixture(L, fx)
    L.set_tap(True)
    nSteps = int(fx["cfg"][4])
    for k in range(1, nSteps + 1):
        sk = "s%d_" % k
        if sk + "rng" in fx:
            assert np.array_equal(L.get_rng_state(), fx[sk + "rng"]), "rng stream diverged before step %d" % k
            sca = L.scalars()
            assert sca.beta == fx[sk + "beta"][0] and sca.CmaxRet == fx[sk + "cmax"][0]
        L.step(1)
        if sk + "flat" in fx:
            assert np.array_equal(L.readback(capi.TAP_FLAT), fx[sk + "flat"])
            assert np.array_equal(L.readback(capi.TAP_TAG), fx[sk + "tag"])
            assert np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"])
            assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"]) < 1e-6
            assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"]) < 1e-6
            assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"]) < 1e-6
            assert relinf(L.readback(capi.TAP_DELTAQ), fx[sk + "dq"]) < 1e-6
            assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"]) < 1e-6
            assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"])
        if sk + "gradSum" in fx or sk + "gradSum_sub" in fx:
            assert fx_vec_dev(fx, sk + "gradSum", L.readback(capi.TAP_GRADSUM)) < 1e-5
        if sk + "W" in fx or sk + "W_sub" in fx:
            w, m1, m2 = L.get_params()
            assert fx_vec_dev(fx, sk + "W", w) < 1e-6
            assert fx_vec_dev(fx, sk + "M1", m1) < 1e-5 and fx_vec_dev(fx, sk + "M2", m2) < 1e-5
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-14 * abs(sca.beta)
        assert sca.CmaxRet == fx["traj_cmax"][k - 1]
        assert sca.nFarPolicySteps == fx["traj_nfar"][k - 1]
        e = fixture_arrival(fx, k)
        if e is not None:
            L.append_episode(**synth_episode(fixture_synth(fx), e, getattr(L, "nOptions", 0)))
    w, _, _ = L.get_params()
    assert fx_vec_dev(fx, "Wfinal", w) < 1e-6


CONV_FIXTURES = ["conv_small.bin", "racer_atari.bin", "appended_dense.bin", "nature_dqn.bin", "lstm_appended.bin", "conv_lstm.bin", "conv_extra.bin", "conv_extra_appended.bin"]      # nature_dqn: the 32 / 64 / 64-channel stack of Builder.cpp:189-194
CONV_FUNC = {"conv_small.bin": "Tanh", "racer_atari.bin": "Tanh", "appended_dense.bin": "SoftSign", "nature_dqn.bin": "Tanh", "lstm_appended.bin": "Tanh", "conv_lstm.bin": "Tanh", "conv_extra.bin": "Tanh", "conv_extra_appended.bin": "Tanh"}      # (settings/RACER_atari.json leaves nnFunc at its default)


@pytest.mark.parametrize("name", CONV_FIXTURES)
def test_conv_layout_init_and_initialize_match_reference(name):
    """Convolutional preprocessing (Approximator::buildPreprocessing -> Builder::addConv2d -> Conv2DLayer): blob layout with one
    bias per output element, Conv2DLayer::initialize draw order, checkpoint packing, then initializeLearner on states of
    (1 + nAppendedObs) frames."""
    fx = load_fixture(name)
    L = oracle_learner(fixture_config(fx, nnFunc=CONV_FUNC[name], episode_order=capi.ORDER_REFERENCE))
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    assert np.array_equal(lay["nW"], fx["nWeights"]) and np.array_equal(lay["nB"], fx["nBiases"])
    L.init_weights()
    w0 = L.get_params()[0]
    assert fx_vec_dev(fx, "W0", w0) < 1e-12 and ("W0" in fx or np.array_equal(w0[::53], fx["W0_sub"]))   # (sums: summation order only)
    assert np.array_equal(L.get_rng_state(), fx["rng_before_init"])
    L = oracle_learner(fixture_config(fx, nnFunc=CONV_FUNC[name], episode_order=capi.ORDER_REFERENCE))
    setup_from_fixture(L, fx)
    s = L.scalars()
    assert s.nStoredSteps == int(fx["cfg"][7]) and s.beta == fx["beta0"][0] and s.CmaxRet == fx["cmax0"][0]
    m, sc, r = L.get_scaling()
    assert np.array_equal(np.concatenate([m, sc, r]), fx["scaling0"])
    assert np.array_equal(L.get_rng_state(), fx["rng0"])


@pytest.mark.parametrize("name", CONV_FIXTURES)
def test_conv_steps_match_reference(name, tmp_path):
    """Steps on the (episode, t >= nAppendedObs) pairs the harness drew (oracle/ref_driver.cpp: RestrictedSampler): stacked
    standardised frames -> SoftSign convolutions -> dense + parametric residual over the first conv outputs -> discrete RACER
    head; outputs, rho, D_KL, gradients, Adam update against the compiled reference."""
    fx = load_fixture(name)
    L = oracle_learner(fixture_config(fx, nnFunc=CONV_FUNC[name], episode_order=capi.ORDER_REFERENCE))
    setup_from_fixture(L, fx)
    L.set_tap(True)
    nSteps = int(fx["cfg"][4])
    for k in range(1, nSteps + 1):
        sk = "s%d_" % k
        assert np.array_equal(L.get_rng_state(), fx[sk + "rng"]), k      # only the Adam draws advance the learner's generator
        flat = flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        assert np.all(np.diff(flat) > 0)
        L.step(1, flat=flat)
        assert np.array_equal(L.readback(capi.TAP_TAG), fx[sk + "tag"]) and np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"]) < 2e-6
        assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"]) < 2e-6
        assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"]) < 2e-6
        assert relinf(L.readback(capi.TAP_DELTAQ), fx[sk + "dq"]) < 2e-6
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"]) < 2e-6
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"])
        if sk + "gradSum" in fx or sk + "gradSum_sub" in fx:
            assert fx_vec_dev(fx, sk + "gradSum", L.readback(capi.TAP_GRADSUM)) < 1e-5
        if sk + "W" in fx or sk + "W_sub" in fx:
            w, m1, m2 = L.get_params()
            assert fx_vec_dev(fx, sk + "W", w) < 1e-6
            assert fx_vec_dev(fx, sk + "M1", m1) < 1e-5 and fx_vec_dev(fx, sk + "M2", m2) < 1e-5
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-14 * abs(sca.beta)
        assert sca.nFarPolicySteps == fx["traj_nfar"][k - 1]
    w, m1, m2 = L.get_params()
    assert fx_vec_dev(fx, "Wfinal", w) < 1e-6
    if "ckpt_net_weights" in fx:      # Conv2DLayer::save (Layer_Conv2D.h:215-231): filters, then biases, as they lie
        ref = np.frombuffer(bytes(bytearray(fx["ckpt_net_weights"])), np.float32)
        base = str(tmp_path / "agent_00_net")
        L.save(base)
        mine = np.fromfile(base + "_weights.raw", np.float32)
        assert mine.size == ref.size and relinf(mine, ref) < 1e-6
        L2 = oracle_learner(fixture_config(fx, nnFunc=CONV_FUNC[name]))
        open(str(tmp_path / "ref_net_weights.raw"), "wb").write(ref.tobytes())
        L2.init_weights(); L2.restart(str(tmp_path / "ref_net")); L2.save(str(tmp_path / "again"))
        assert np.array_equal(np.fromfile(str(tmp_path / "again_weights.raw"), np.float32), ref)


def test_importance_weight_histogram_matches_reference_printout():
    """MemoryProcessing::histogramImportanceWeights (MemoryProcessing.cpp:353-389) after 40 steps: the block the compiled
    reference printed, character for character."""
    fx, L = make("hist_small.bin")
    setup_from_fixture(L, fx)
    L.step(40)
    assert relinf(L.get_params()[0], fx["Wfinal"]) < 1e-6
    text, cnt = L.impweight_histogram()
    ref = bytes(bytearray(fx["impw_histogram"])).decode().rstrip("\n")
    assert cnt.sum() == L.scalars().nStoredSteps and cnt[0] > 0 and (cnt[40:60] > 0).any()
    assert text == ref


def test_far_policy_masks_are_exercised():
    """The fixtures must contain both accepted and rejected (far-policy) samples."""
    fx = load_fixture("small_mixed.bin")
    far = np.concatenate([fx["s%d_far" % k] for k in range(1, 13)])
    assert 0 < far.sum() < far.size


@pytest.mark.parametrize("name", ["traj_1200.bin", "racer_traj_1200.bin", "moving_traj_1200.bin"])
def test_long_trajectory_crosses_1000_step_sweep(tmp_path, name):
    """1200 steps: beta / CmaxRet / nFarPolicySteps trajectories, the 1000-step
    Episode::updateCumulative + full Retrace sweep and the reward/state statistics EMA.  moving_traj_1200: with an episode
    arriving behind every third step and the oldest ones leaving (400 arrivals, the replay turned over several times)."""
    fx, L = make(name)
    setup_from_fixture(L, fx)
    L.set_log_base(str(tmp_path / "agent_00"))
    lens = {e: synth_episode(fixture_synth(fx), e)["rewards"].size for e in range(int(fx["cfg"][3]) + 400)}
    for k in range(1, 1201):
        L.step(1)
        if fixture_arrival(fx, k) is not None:          # (the recording run appended it right behind the step, before its dumps)
            L.append_episode(**synth_episode(fixture_synth(fx), fixture_arrival(fx, k)))
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-12 * abs(sca.beta), k
        assert sca.nFarPolicySteps == fx["traj_nfar"][k - 1], k
        assert sca.CmaxRet == fx["traj_cmax"][k - 1]
        sk = "s%d_" % k
        if sk + "ret" in fx:
            for field, key, tol in ((capi.EP_RETURN, "ret", 2e-5), (capi.EP_VALUE, "val", 2e-5),
                                    (capi.EP_IMPW, "impw", 2e-5), (capi.EP_DKL, "ep_dkl", 2e-5),
                                    (capi.EP_DELTAQ, "ep_dq", 2e-4)):
                mine = episode_arrays_by_tag(L, field)
                ref = fixture_arrays_by_tag(fx, sk + "ep_tags", sk + key, lens)
                for tag, arr in ref.items():
                    assert np.allclose(mine[tag], arr, rtol=tol, atol=tol), (k, key, tag)
            m, sc, r = L.get_scaling()
            assert np.allclose(np.concatenate([m, sc, r]), fx[sk + "scaling"], rtol=1e-6, atol=1e-7)
    w, _, _ = L.get_params()
    assert relinf(w, fx["Wfinal"]) < 1e-5
    st = L.stats()
    ref = fx["stats_final"]
    mine = [st.avgKLdivergence, st.avgSquaredErr, st.maxAbsError, st.avgReturn, st.avgQ, st.stdevQ, st.minQ, st.maxQ]
    assert np.allclose(mine, ref[:8], rtol=1e-4, atol=1e-6)
    assert st.nFarPolicySteps == int(ref[8])
    # StatsTracker: the output-gradient statistics file the reference wrote at steps 0 and 1000, and the
    # mean / RMS over its last minibatch
    ref_file = np.frombuffer(bytes(bytearray(fx["outgrad_stats_file"])), np.float32)
    mine_file = np.fromfile(str(tmp_path / "agent_00_net_outGrad_stats.raw"), np.float32)
    assert mine_file.size == ref_file.size == 1 + 2 * 2 * L.nOut and mine_file[0] == ref_file[0]
    assert np.allclose(mine_file, ref_file, rtol=2e-5, atol=1e-7)
    m, r = L.grad_stats()
    assert np.allclose(np.concatenate([m, r]), fx["outgrad_stats_last"], rtol=2e-5, atol=1e-7)
    # the statistics line of agent_00_stats.txt for this state, as the reference itself printed it
    head = bytes(bytearray(fx["metrics_head"])).decode()
    assert lines_agree(stats_line(L), bytes(bytearray(fx["metrics_line"])).decode(), head)


@pytest.mark.parametrize("name", ["small_mixed.bin", "deep_tanh.bin", "racer_lstm.bin", "vracer_mgu.bin", "racer_discrete.bin"])
def test_checkpoint_files_match_reference(name, tmp_path):
    """ol_save writes byte-for-byte the files Approximator::save wrote for the same weights and
    moments (Network.cpp:22-38); ol_restart of the reference's files restores the padded blobs."""
    fx = load_fixture(name)
    L = oracle_learner(fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    L.set_params(fx["Wfinal"], fx["M1final"], fx["M2final"])
    base = str(tmp_path / "agent_00_net")
    L.save(base)
    for suf in ("_weights", "_1stMom", "_2ndMom"):
        assert open(base + suf + ".raw", "rb").read() == bytes(bytearray(fx["ckpt_net" + suf])), suf
    ref = str(tmp_path / "ref_net")
    for suf in ("_weights", "_1stMom", "_2ndMom"):
        open(ref + suf + ".raw", "wb").write(bytes(bytearray(fx["ckpt_net" + suf])))
    L2 = oracle_learner(fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    L2.init_weights()
    L2.restart(ref)
    w, m1, m2 = L2.get_params()
    assert np.array_equal(w, fx["Wfinal"]) and np.array_equal(m1, fx["M1final"]) and np.array_equal(m2, fx["M2final"])
    with pytest.raises(Exception):
        L2.restart(str(tmp_path / "missing"))


@pytest.mark.parametrize("name", ["small_mixed.bin", "racer_discrete.bin"])
def test_packed_episode_wire_format_matches_reference(name):
    from parity import check_packed_roundtrip
    check_packed_roundtrip(oracle_learner, load_fixture(name))


def test_episode_log_of_a_replay_in_motion_matches_reference(tmp_path):
    """cumulative_rewards.dat (MemoryBuffer::pushBackEpisode, MemoryBuffer.cpp:491-513; --logAllSamples): one line
    "nGradSteps timeStamp agentID nSteps totalReward" per episode entering the training set -- the file the compiled reference wrote
    while moving_replay.bin was recorded (25 episodes before training, 30 behind every second gradient step: gradient-step counts and
    time stamps from minTotObsNum observations on) against the oracle's, line by line.  (The harness uses the agent id as the
    episode's content tag; the library's callers log agent 0.)"""
    fx, L = make("moving_replay.bin")
    L.set_episode_log(tmp_path / "rewards.dat")
    setup_from_fixture(L, fx)
    for k in range(1, int(fx["cfg"][4]) + 1):
        L.step(1)
        e = fixture_arrival(fx, k)
        if e is not None:
            L.append_episode(**synth_episode(fixture_synth(fx), e))
    ref = bytes(fx["rewards_log"]).decode().splitlines()
    mine = open(tmp_path / "rewards.dat").read().splitlines()
    assert len(ref) == len(mine) == 55
    for a, b in zip(ref, mine):
        a, b = a.split(), b.split()
        assert [a[0], a[1], a[3], a[4]] == [b[0], b[1], b[3], b[4]], (a, b)
    assert ref[-1].split()[1] == "910"
