"""CPU suite, round 3: the rest of the settings surface of the path -- returnsEstimator (GAE, retraceExplore, none), nnOutputFunc,
encoderLayerSizes, nnType "RNN" -- and the dRet column of the statistics line.  The oracle (oracle/port) against fixtures the
compiled reference recorded with those settings (tests/golden/make_golden.sh, round-3 section)."""
import numpy as np
import pytest

from oracle_api import oracle_learner, synth_episode
from parity import (load_fixture, fixture_config, fixture_synth, fixture_arrival, setup_from_fixture, relinf,
                    episode_arrays_by_tag, fixture_arrays_by_tag, lines_agree, stats_file_lines)
from smarties_amd import capi

FUNC_OF = {"outfunc_lrelu_gauss.bin": "Tanh", "encoder_dense.bin": "Tanh", "vracer_rnn.bin": "Tanh"}
SHORT = ["ret_none.bin", "outfunc_tanh.bin", "outfunc_lrelu_gauss.bin", "outfunc_sigm_discrete.bin", "encoder_dense.bin",
         "vracer_rnn.bin", "discrete_rnn.bin"]
LONG = ["ret_gae.bin", "ret_explore.bin", "stats_2100.bin"]


def make(name, **over):
    fx = load_fixture(name)
    L = oracle_learner(fixture_config(fx, nnFunc=FUNC_OF.get(name), episode_order=capi.ORDER_REFERENCE, **over))
    return fx, L


def test_fixtures_carry_the_settings_they_pin():
    assert int(load_fixture("ret_gae.bin")["retEst"][0]) == capi.RET["GAE"]
    assert int(load_fixture("ret_explore.bin")["retEst"][0]) == capi.RET["retraceExplore"]
    assert int(load_fixture("ret_none.bin")["retEst"][0]) == capi.RET["none"]
    assert int(load_fixture("outfunc_tanh.bin")["outFunc"][0]) == capi.FUNC["Tanh"]
    assert int(load_fixture("outfunc_lrelu_gauss.bin")["outFunc"][0]) == capi.FUNC["LRelu"]
    assert int(load_fixture("outfunc_sigm_discrete.bin")["outFunc"][0]) == capi.FUNC["Sigm"]
    assert [int(x) for x in load_fixture("encoder_dense.bin")["encoder"]] == [24, 0]    # (createEncoder drops the zero entry)
    assert int(load_fixture("vracer_rnn.bin")["cfg"][13]) == capi.NN_RNN


@pytest.mark.parametrize("name", SHORT + LONG)
def test_layout_init_and_rng_match_reference(name):
    """Blob layout (recurrent dense layers hold [W_in; W_rec], Layer_Base.h:24-28), Layer::initialize draw order, the initial
    output biases as pre-images under nnOutputFunc (Layer_Base.h:122-125): weights and generator state bit-exact."""
    fx, L = make(name)
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    assert np.array_equal(lay["nW"], fx["nWeights"]) and np.array_equal(lay["nB"], fx["nBiases"])
    L.init_weights()
    assert np.array_equal(L.get_params()[0], fx["W0"])
    assert np.array_equal(L.get_rng_state(), fx["rng_before_init"])


def test_encoder_layers_are_the_first_hidden_layers():
    """encoderLayerSizes [24, 0] + nnLayerSizes [16, 16] == nnLayerSizes [24, 16, 16] (one network, Learner_approximator.cpp:149-166)"""
    fx, A = make("encoder_dense.bin")
    _, Bq = make("encoder_dense.bin", encoder=[], hidden=[24, 16, 16])
    A.init_weights(); Bq.init_weights()
    assert np.array_equal(A.get_params()[0], Bq.get_params()[0])


@pytest.mark.parametrize("name", SHORT + LONG)
def test_initialize_matches_reference(name):
    """Learner::initializeLearner: rescaleAllReturnEstimator with the configured estimator (none: estimates stay zero)."""
    fx, L = make(name)
    setup_from_fixture(L, fx)
    s = L.scalars()
    assert s.beta == fx["beta0"][0] and s.CmaxRet == fx["cmax0"][0]
    m, sc, r = L.get_scaling()
    assert np.array_equal(np.concatenate([m, sc, r]), fx["scaling0"])
    nopt = getattr(L, "nOptions", 0)
    lens = {e: synth_episode(fixture_synth(fx), e, nopt)["rewards"].size for e in range(int(fx["cfg"][3]))}
    mine = episode_arrays_by_tag(L, capi.EP_RETURN)
    ref = fixture_arrays_by_tag(fx, "ret0_tags", "ret0", lens)
    for tag, arr in ref.items():
        assert np.allclose(mine[tag], arr, rtol=1e-6, atol=1e-6), tag
    if name == "ret_none.bin":
        assert not any(a.any() for a in mine.values())
    assert np.array_equal(L.get_rng_state(), fx["rng0"])


@pytest.mark.parametrize("name", SHORT)
def test_steps_match_reference(name):
    fx, L = make(name)
    setup_from_fixture(L, fx)
    L.set_tap(True)
    nopt = getattr(L, "nOptions", 0)
    lens = {e: synth_episode(fixture_synth(fx), e, nopt)["rewards"].size for e in range(int(fx["cfg"][3]))}
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        if sk + "rng" in fx:
            assert np.array_equal(L.get_rng_state(), fx[sk + "rng"]), "rng stream diverged before step %d" % k
        L.step(1)
        if sk + "flat" in fx:
            assert np.array_equal(L.readback(capi.TAP_FLAT), fx[sk + "flat"])
            assert np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"])
            assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"]) < 1e-6
            assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"]) < 1e-6
            assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"]) < 1e-6
            assert relinf(L.readback(capi.TAP_DELTAQ), fx[sk + "dq"]) < 1e-6
            assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"]) < 1e-6
            assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"])
        if sk + "gradSum" in fx:
            assert relinf(L.readback(capi.TAP_GRADSUM), fx[sk + "gradSum"]) < 1e-5
        if sk + "W" in fx:
            w, m1, m2 = L.get_params()
            assert relinf(w, fx[sk + "W"]) < 1e-6
            assert relinf(m1, fx[sk + "M1"]) < 1e-5 and relinf(m2, fx[sk + "M2"]) < 1e-5
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-14 * abs(sca.beta)
        assert sca.nFarPolicySteps == fx["traj_nfar"][k - 1]
        if sk + "ret" in fx:
            mine = episode_arrays_by_tag(L, capi.EP_RETURN)
            for tag, arr in fixture_arrays_by_tag(fx, sk + "ep_tags", sk + "ret", lens).items():
                assert np.allclose(mine[tag], arr, rtol=2e-6, atol=2e-6), (k, tag)
    assert relinf(L.get_params()[0], fx["Wfinal"]) < 1e-6
    head, line = L.metrics()         # no sweep so far: no dRet column, as in the reference's own line
    assert head == bytes(bytearray(fx["metrics_head"])).decode()
    assert lines_agree(line, bytes(bytearray(fx["metrics_line"])).decode(), head)


@pytest.mark.parametrize("name", LONG)
def test_trajectories_and_statistics_lines_across_the_sweeps(name):
    """beta / far-policy trajectories with GAE / retraceExplore (its bonus reads ReplayStats::maxAbsError as of before the
    step's update) across the 1000-step sweeps, the estimates right after a sweep, and the lines the reference wrote into
    <learner>_stats.txt at steps 1000 (2000): the dRet column is printed when a sweep ran since the last line, and
    printing it consumes the counters (MemoryBuffer.cpp:534-544)."""
    fx, L = make(name)
    setup_from_fixture(L, fx)
    nSteps = int(fx["cfg"][4])
    lens = {e: synth_episode(fixture_synth(fx), e)["rewards"].size for e in range(int(fx["cfg"][3]) + nSteps)}
    ref_head, ref_lines = stats_file_lines(fx)
    assert "dRet" in ref_head and len(ref_lines) == nSteps // 1000
    for k in range(1, nSteps + 1):
        L.step(1)
        if fixture_arrival(fx, k) is not None:
            L.append_episode(**synth_episode(fixture_synth(fx), fixture_arrival(fx, k)))
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-12 * abs(sca.beta), k
        assert sca.nFarPolicySteps == fx["traj_nfar"][k - 1], k
        sk = "s%d_" % k
        if sk + "ret" in fx:
            mine = episode_arrays_by_tag(L, capi.EP_RETURN)
            for tag, arr in fixture_arrays_by_tag(fx, sk + "ep_tags", sk + "ret", lens).items():
                assert np.allclose(mine[tag], arr, rtol=2e-5, atol=2e-5), (k, tag)
        if k % 1000 == 0:            # Learner::logStats at currStep % freqPrint == 0 (Learner.cpp:134-156)
            if k == 1000:
                st = L.stats()
                assert st.countReturnsEstimateUpdates > 0 and st.sumReturnsEstimateErrors > 0
            head, line = L.metrics()
            assert head == ref_head
            # every column but the last: the reference prints before this step's Adam update, so its weight norm is one update older
            ref = ref_lines[k // 1000 - 1]
            assert len(line) == len(ref)
            assert lines_agree(" ".join(line.split()[:-1]), " ".join(ref.split()[:-1]), head, rel=2e-5), (line, ref)
            assert abs(float(line.split()[-1]) - float(ref.split()[-1])) < 1e-2
            assert L.stats().countReturnsEstimateUpdates == 0
    assert relinf(L.get_params()[0], fx["Wfinal"]) < 1e-5
    head, line = L.metrics()         # between two sweeps: the column is gone again
    assert head == bytes(bytearray(fx["metrics_head"])).decode() and "dRet" not in head
    assert lines_agree(line, bytes(bytearray(fx["metrics_line"])).decode(), head, rel=2e-5)
    assert L.stats().countReturnsEstimateUpdates == -1


@pytest.mark.parametrize("name", ["encoder_dense.bin", "vracer_rnn.bin"])
def test_checkpoint_files_match_reference(name, tmp_path):
    fx, L = make(name)
    setup_from_fixture(L, fx)
    L.step(int(fx["cfg"][4]))
    base = str(tmp_path / "ck_net")
    L.save(base)
    for suf in ("_weights", "_1stMom", "_2ndMom"):
        mine = np.fromfile(base + suf + ".raw", np.float32)
        ref = np.frombuffer(bytes(bytearray(fx["ckpt_net" + suf])), np.float32)
        assert mine.size == ref.size
        assert relinf(mine, ref) < 2e-5
    # the reference's own files restore the padded blobs
    for suf in ("_weights", "_1stMom", "_2ndMom"):
        np.frombuffer(bytes(bytearray(fx["ckpt_net" + suf])), np.float32).tofile(base + suf + ".raw")
    M = oracle_learner(fixture_config(fx, nnFunc=FUNC_OF.get(name)))
    M.init_weights(); M.restart(base)
    for a, b in zip(M.get_params(), (fx["Wfinal"], fx["M1final"], fx["M2final"])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["small_mixed.bin", "ns_shape.bin", "hp_lowclip.bin", "threads3.bin", "hp_odd.bin"])
def test_far_policy_count_is_the_reference_loop_over_the_storage_order(name):
    """ReplayStats::nFarPolicySteps is `Uint += float * float` per episode in storage order (MemoryProcessing.cpp:202-238): the helper
    the GPU suite uses (parity.far_count_loop: exact fused multiply-add, x86's float -> unsigned conversion) reproduces (i) the
    fixture's number from the fractions of a learner in the OTHER (newest-first) storage order, walked in the reference's order, and
    (ii) that learner's own number over its own order -- at every step."""
    from parity import far_count_loop, storage_order, flat_for
    fx = load_fixture(name)
    R = oracle_learner(fixture_config(fx, episode_order=capi.ORDER_REFERENCE)); setup_from_fixture(R, fx)
    S = oracle_learner(fixture_config(fx)); setup_from_fixture(S, fx)
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        if sk + "flat" not in fx:
            break
        ref_order = storage_order(R)                   # (a step's statistics pass runs before its std::sort)
        R.step(1)
        S.step(1, flat=np.sort(flat_for(S, fx[sk + "tag"], fx[sk + "t"])))
        assert R.scalars().nFarPolicySteps == fx["traj_nfar"][k - 1]
        assert far_count_loop(S, ref_order) == fx["traj_nfar"][k - 1], k
        assert S.scalars().nFarPolicySteps == far_count_loop(S, storage_order(S)), k
