"""Network combinations of Approximator::buildPreprocessing / buildFromSettings (Network/Approximator.cpp:218-271,
Network/Builder.cpp:26-46,76-81) beyond the shipped settings files: recurrent layers behind appended observations, recurrent layers
wider than 64 cells, (further down) state variables beside the image and recurrent layers behind a convolutional stack.
Each against a fixture recorded from the compiled reference (tests/golden/make_golden.sh, "G-a22") and against the oracle on
minibatches the library draws itself."""
import numpy as np
import pytest

from smarties_amd import capi
from oracle_api import synth_cfg
from parity import load_fixture, fixture_config, setup_from_fixture, relinf, fx_vec_dev, flat_for
from test_hip_parity import hip_learner, _pair, _compare_step, TOL32

pytestmark = pytest.mark.gpu

A22_FIXTURES = {  # name -> (nnFunc, minibatches drawn by the harness' restricted sampler?)
    "lstm_wide.bin": ("Tanh", False), "mgu_wide.bin": ("Tanh", False), "lstm_appended.bin": ("Tanh", True),
}


@pytest.mark.parametrize("name", sorted(A22_FIXTURES))
def test_steps_follow_reference_fixture_a22(hip_api, name):
    """The (episode, t) pairs of every tapped step of the reference run, fed to the library: per-sample outputs, importance weights,
    gradients; summed gradient, weights and Adam moments where the fixture holds them (lean fixtures: every 53rd element + sums)."""
    func, restricted = A22_FIXTURES[name]
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc=func))
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    setup_from_fixture(L, fx)
    w0 = L.get_params()[0]
    assert fx_vec_dev(fx, "W0", w0) < 1e-12 and ("W0" in fx or np.array_equal(w0[::53], fx["W0_sub"]))
    assert np.array_equal(L.get_rng_state(), fx["rng0"])
    nSteps = int(fx["cfg"][4])
    for k in range(1, nSteps + 1):
        sk = "s%d_" % k
        if sk + "tag" not in fx:
            break
        flat = flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        order = np.argsort(flat, kind="stable")
        L.step(1, flat=flat[order])
        assert np.array_equal(L.readback(capi.TAP_TAG), fx[sk + "tag"][order])
        assert np.array_equal(L.readback(capi.TAP_TSTEP), fx[sk + "t"][order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_DKL), fx[sk + "dkl"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_DELTAQ), fx[sk + "dq"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"][order]) < TOL32
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"][order])
        if sk + "gradSum" in fx or sk + "gradSum_sub" in fx:
            assert fx_vec_dev(fx, sk + "gradSum", L.readback(capi.TAP_GRADSUM)) < TOL32
        if sk + "W" in fx or sk + "W_sub" in fx:
            w, m1, m2 = L.get_params()
            assert fx_vec_dev(fx, sk + "W", w) < TOL32
            assert fx_vec_dev(fx, sk + "M1", m1) < TOL32 and fx_vec_dev(fx, sk + "M2", m2) < 2 * TOL32
        sca = L.scalars()
        assert abs(sca.beta - fx["traj_beta"][k - 1]) <= 1e-12 * abs(sca.beta)
    assert fx_vec_dev(fx, "Wfinal", L.get_params()[0]) < 2 * TOL32


WIDE_SHAPES = [  # (layer type, hidden, dimS, nAppendedObs, bptt, batch)
    ("lstm", (96, 80), 5, 0, 6, 16),        # four gates x 96 cells: gates looped over the 256 threads
    ("lstm", (256,), 7, 0, 3, 6),           # the widest: 1024 gates, weights read through the L2
    ("lstm", (96, 80, 64), 6, 0, 5, 20),    # three layers time-step-major: the middle layer's deltas have both their producers on one diagonal
    ("lstm", (128, 128), 5, 0, 16, 40),     # the window the shipped preset uses, samples in three blocks of 16 (one partial)
    ("lstm", (512,), 7, 0, 3, 6),           # beyond the per-sample kernels' 256 cells (round 5): training AND acting windows time-step-major
    ("lstm", (320, 272), 9, 2, 4, 12),      # ... with appended observations in front
    ("mgu", (384, 320), 6, 0, 3, 10),
    ("mgu", (128,), 6, 0, 5, 16),
    ("mgu", (200, 72), 9, 0, 4, 9),
    ("mgu", (96, 80), 5, 0, 6, 16),         # time-step-major launches, reductions that are not whole groups of 16 per wavefront
    ("mgu", (256, 128), 7, 0, 3, 20),       # ... a partial block of 16 samples
    ("mgu", (96, 80, 64), 6, 0, 5, 20),     # three layers: the middle layer's deltas have both their producers on one diagonal
    ("rnn", (130, 70), 5, 0, 5, 12),        # dense layers with a recurrent term, more than 64 cells
    ("lstm", (32, 32), 5, 2, 4, 16),        # appended observations in front of the shipped recurrent shape
    ("lstm", (24,), 40, 7, 6, 10),          # 320 inputs
    ("mgu", (32, 32), 4, 3, 16, 33),
    ("rnn", (24, 16), 6, 1, 5, 12),
    ("lstm", (80,), 11, 2, 5, 8),           # both
]


@pytest.mark.parametrize("shape", WIDE_SHAPES, ids=lambda sh: "%s-%s-dS%d-app%d" % (sh[0], "x".join(map(str, sh[1])), sh[2], sh[3]))
def test_wide_and_appended_recurrent_layers_match_oracle(hip_api, shape):
    """Minibatches the library draws itself -- including steps t < nAppendedObs and windows that begin before them, where the
    appended slots repeat the episode's first state -- eager and replayed steps, then acting on windows of every length with and
    without context states in front."""
    kind, hidden, dS, nApp, bptt, batch = shape
    nnt = {"lstm": capi.NN_LSTM, "mgu": capi.NN_MGU, "rnn": capi.NN_RNN}[kind]
    kw = dict(dimS=dS, dimA=2, bounded=[1, 0], hidden=hidden, nnFunc="Tanh", batchSize=batch, maxTotObsNum=8000, randSeed=5,
              nn_type=nnt, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=bptt, nAppendedObs=nApp)
    G, O = _pair(hip_api, kw, synth_cfg(seed=21, dimS=dS, dimA=2, lenMin=2, lenMax=30, pTerm=0.5), 60)
    for _ in range(3):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(10); O.step(10)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    rng = np.random.default_rng(5)
    for n in sorted({1, 2, bptt + 1, bptt + 1 + nApp}):
        S = rng.normal(size=(n, dS)).astype(np.float32)
        assert relinf(G.forward_sequence(S), O.forward_sequence(S)) < TOL32, n
    with pytest.raises(capi.HlError):
        G.forward_sequence(rng.normal(size=(bptt + 2 + nApp, dS)).astype(np.float32))


from test_hip_parity import test_conv_steps_follow_reference_fixture as _conv_fixture_body
from test_hip_parity import test_conv_and_appended_observations_match_oracle as _conv_oracle_body


@pytest.mark.parametrize("name", ["conv_extra.bin", "conv_extra_appended.bin"])
def test_state_variables_beside_the_image_follow_reference_fixture(hip_api, name):
    """A state wider than the first convolution's image: the surplus is a second input layer behind the conv stack, glued IN FRONT of
    its outputs (Approximator.cpp:249-259, Builder.cpp:26-46, JoinLayer::forward); with appended observations the image is simply the
    first entries of the stacked vector.  Same checks as for the other convolutional fixtures."""
    _conv_fixture_body(hip_api, name)


@pytest.mark.parametrize("cfg_kw,sc_kw,n_eps,steps", [
    # 6 extra variables behind two convolutions, no appended observations; short episodes: truncated next states are sampled
    (dict(dimS=1030, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=5, conv=[(8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)],
          hidden=(48,), nnFunc="Tanh", batchSize=24, maxTotObsNum=2000, randSeed=3),
     dict(seed=5, dimS=1030, dimA=1, lenMin=3, lenMax=9, pTerm=0.3), 60, 6),
    # one extra variable (every offset behind it unaligned), continuous head, two dense layers, a strided first layer
    (dict(dimS=801, dimA=2, bounded=[1, 0], nAppendedObs=3, conv=[(20, 20, 8, 16, 6, 2)], hidden=(40, 24), batchSize=16,
          maxTotObsNum=1500, randSeed=4),
     dict(seed=6, dimS=801, dimA=2, lenMin=4, lenMax=20, pTerm=0.5), 30, 5),
    # more extras than the dense layer is wide (the parametric residual reads extras only)
    (dict(dimS=229, dimA=2, nAppendedObs=0, conv=[(9, 7, 3, 5, 3, 1)], hidden=(32, 32), nnFunc="SoftSign", batchSize=10,
          maxTotObsNum=800, randSeed=6),
     dict(seed=8, dimS=229, dimA=2, lenMin=3, lenMax=12, pTerm=0.4), 25, 5),
], ids=["6-extras", "1-extra-appended", "40-extras"])
def test_state_variables_beside_the_image_match_oracle(hip_api, cfg_kw, sc_kw, n_eps, steps):
    _conv_oracle_body(hip_api, cfg_kw, sc_kw, n_eps, steps)


def test_recurrent_layers_behind_convolutions_follow_reference_fixture(hip_api):
    """conv_lstm.bin: an LSTM layer behind two convolutions on 1 + 3 stacked frames, BPTT 4: every step of a sample's window passes
    through the conv stack (rows = B x 5 window rows + next states), the filter gradients sum over all of them."""
    _conv_fixture_body(hip_api, "conv_lstm.bin")


@pytest.mark.parametrize("kind,hidden,extra", [("lstm", (32,), 0), ("mgu", (24, 16), 0), ("rnn", (20,), 0), ("lstm", (16, 16), 6)],
                         ids=["lstm-32", "mgu-24x16", "rnn-20", "lstm-2x16-extras"])
def test_recurrent_layers_behind_convolutions_match_oracle(hip_api, kind, hidden, extra):
    """Minibatches the library draws itself (short episodes: windows shorter than the BPTT length, steps t < nAppendedObs, truncated
    next states), eager steps only for these nets; then acting on windows of every length, with context states in front."""
    nnt = {"lstm": capi.NN_LSTM, "mgu": capi.NN_MGU, "rnn": capi.NN_RNN}[kind]
    dS = 256 + (extra and 2)          # (with 1 + 2 stacked observations: 3 x 258 = 768 + 6)
    nApp = 2 if extra else 3
    conv = [(8, 8, 12, 32, 4, 1), (5, 5, 32, 64, 3, 1)] if extra else [(8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]
    kw = dict(dimS=dS, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=5, nAppendedObs=nApp, conv=conv, hidden=hidden, nnFunc="Tanh",
              batchSize=12, maxTotObsNum=2000, randSeed=3, nn_type=nnt, nnBPTTseq=4)
    G, O = _pair(hip_api, kw, synth_cfg(seed=5, dimS=dS, dimA=1, lenMin=3, lenMax=12, pTerm=0.3), 50)
    for _ in range(4):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(9); O.step(9)
    _compare_step(G, O)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    rng = np.random.default_rng(7)
    for n in (1, 3, 5, 5 + nApp):
        S = rng.normal(size=(n, dS)).astype(np.float32)
        assert relinf(G.forward_sequence(S), O.forward_sequence(S)) < TOL32, n


def test_rnn_encoder_under_mgu_layers_follows_reference_fixture(hip_api):
    """pomdp_encoder.bin: a partially observable MDP with nnType left at its default -- Approximator::buildPreprocessing makes the
    encoder layers plain recurrent ones ("RNN", Approximator.cpp:264-270), buildFromSettings the layers behind them MGU (:221-223).
    On the device the stack runs as two window launches each way (rec.hip), the lower one's rows being the upper one's input."""
    name = "pomdp_encoder.bin"
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc="Tanh"))
    assert L.nParams == int(fx["cfg"][5]) and L.nOut == int(fx["cfg"][6])
    lay = L.layout()
    assert np.array_equal(lay["indW"], fx["indWeights"]) and np.array_equal(lay["indB"], fx["indBiases"])
    setup_from_fixture(L, fx)
    w, m1, m2 = L.get_params()
    assert np.array_equal(w, fx["W0"]) and np.array_equal(L.get_rng_state(), fx["rng0"])
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        if sk + "flat" not in fx:
            break
        flat = flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        order = np.argsort(flat, kind="stable")
        L.step(1, flat=flat[order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_RHO), fx[sk + "rho"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"][order]) < TOL32
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"][order])
        if sk + "gradSum" in fx:
            assert relinf(L.readback(capi.TAP_GRADSUM), fx[sk + "gradSum"]) < TOL32
        if sk + "W" in fx:
            w, m1, m2 = L.get_params()
            assert relinf(w, fx[sk + "W"]) < TOL32 and relinf(m1, fx[sk + "M1"]) < TOL32 and relinf(m2, fx[sk + "M2"]) < 2 * TOL32


@pytest.mark.parametrize("enc,hidden,nApp,conv", [((24,), (16, 16), 0, None), ((20, 36), (32,), 2, None), ((16,), (16, 8), 0, [(8, 8, 4, 8, 3, 1)])],
                         ids=["24|16x16", "20x36|32-appended", "conv|16|16x8"])
def test_rnn_encoder_under_mgu_layers_matches_oracle(hip_api, enc, hidden, nApp, conv):
    dS = 256 if conv else 6
    kw = dict(dimS=dS, dimA=2, bounded=[1, 0], hidden=hidden, encoder=list(enc), encoder_rnn=1, nn_type=capi.NN_MGU, nnFunc="Tanh", batchSize=12,
              maxTotObsNum=4000, randSeed=5, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=5, nAppendedObs=nApp)
    if conv:
        kw["conv"] = conv
    G, O = _pair(hip_api, kw, synth_cfg(seed=21, dimS=dS, dimA=2, lenMin=2, lenMax=30, pTerm=0.5), 50)
    for _ in range(3):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(10); O.step(10)
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    rng = np.random.default_rng(5)
    for n in sorted({1, 3, 6, 6 + nApp}):
        S = rng.normal(size=(n, dS)).astype(np.float32)
        assert relinf(G.forward_sequence(S), O.forward_sequence(S)) < TOL32, n
