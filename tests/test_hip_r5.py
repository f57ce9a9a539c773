"""Round 5, on the GPU: the sample-resident convolution kernels (convt.hip) and the LDS-staged filter gradients (conv.hip:
convDwStaged) against the compiled reference's fixtures, in every form the library can run them."""
import numpy as np
import pytest

from smarties_amd import capi
from parity import load_fixture, fixture_config, setup_from_fixture, flat_for, relinf, fx_vec_dev
from test_hip_parity import hip_learner, TOL32


def run_fixture(hip_api, name):
    fx = load_fixture(name)
    L = hip_learner(hip_api, fixture_config(fx, nnFunc="Tanh"))
    setup_from_fixture(L, fx)
    for k in range(1, int(fx["cfg"][4]) + 1):
        sk = "s%d_" % k
        flat = flat_for(L, fx[sk + "tag"], fx[sk + "t"])
        order = np.argsort(flat, kind="stable")
        L.step(1, flat=flat[order])
        assert relinf(L.readback(capi.TAP_OUTPUT), fx[sk + "O"][order]) < TOL32
        assert relinf(L.readback(capi.TAP_OUTGRAD), fx[sk + "G"][order]) < TOL32
        assert np.array_equal(L.readback(capi.TAP_FAR), fx[sk + "far"][order])
        if sk + "gradSum" in fx or sk + "gradSum_sub" in fx:
            assert fx_vec_dev(fx, sk + "gradSum", L.readback(capi.TAP_GRADSUM)) < TOL32
    w = L.get_params()[0]
    assert fx_vec_dev(fx, "Wfinal", w) < TOL32
    L.close()
    return w


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["racer_atari.bin", "conv_small.bin", "nature_dqn.bin"])
def test_every_form_of_the_convolutional_backward_pass_follows_the_reference(hip_api, monkeypatch, name):
    """Layer_Conv2D.h:117-138 through (default) the kernels with the RACER_atari geometry at compile time / the any-geometry
    sample-resident kernel, (GENERIC & 16) the any-geometry kernels on every stack, (& 8) the per-layer launches of conv.hip; filter
    gradients staged in LDS (default) or gathered (& 32); everything general at once (stacked rows, no row blocks, separate launches).  Each follows the reference's taps; among themselves they differ by
    summation order only."""
    ws = {}
    for tag, env in (("default", {}), ("any_geometry", {"SMARTIES_HIP_GENERIC": "16"}), ("per_layer", {"SMARTIES_HIP_GENERIC": "8"}),
                     ("gather_dw", {"SMARTIES_HIP_GENERIC": "32"}), ("all_general", {"SMARTIES_HIP_GENERIC": str(8 + 16 + 32 + 64 + 256)})):
        monkeypatch.delenv("SMARTIES_HIP_GENERIC", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ws[tag] = run_fixture(hip_api, name)
    ref = ws["default"]
    for tag, w in ws.items():
        assert np.abs(w - ref).max() <= 1e-5 * np.abs(ref).max(), tag


@pytest.mark.gpu
@pytest.mark.parametrize("B,nEps", [(4096, 250), (16384, 700)], ids=["b4096", "b16384"])
def test_bench_network_at_the_timed_large_batches_matches_oracle(hip_api, B, nEps):
    """bench.py's other_configs rows cfgNS_2x256_b4096 / b16384 (17 states, 6 bounded actions, 2 x 256 SoftSign) at their own shape:
    the 64 x 64 weight-gradient tiles over row chunks and the weight-stationary panels (bigmm.hip) at K = B rows x N = 256 columns,
    the panel head, the 1024-thread sampler -- sampled indices, generator state, per-sample taps and the updated weights against
    the oracle (these rows were timed, not checked; Learner_approximator.cpp:67-77 is shape-agnostic)."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=B, maxTotObsNum=1048576, randSeed=42)
    G, O = _pair(hip_api, kw, synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3), nEps)
    for _ in range(2):
        G.step(1); O.step(1)
        assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
        _compare_step(G, O)
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
def test_timed_out_gradient_exchange_leaves_the_parameters_as_they_were(hip_api, monkeypatch):
    """One of two connected replicas steps, its peer never does: the exchange kernel's wait ends after SMARTIES_HIP_XCHG_TIMEOUT_MS and
    the learner reports a device error -- with weights and both moments exactly those of before the step (round 5: no chunk applies
    Adam before all chunks' peers have arrived; until round 4 the update could be partial).  Optimizer.cpp:110-132."""
    from oracle_api import synth_cfg
    from test_hip_parity import _xchg_replicas, _both
    monkeypatch.setenv("SMARTIES_HIP_XCHG_TIMEOUT_MS", "300")
    cfg_kw = dict(dimS=6, dimA=2, hidden=(64, 64), batchSize=32, maxTotObsNum=4000, randSeed=5)
    Ls = _xchg_replicas(hip_api, cfg_kw, synth_cfg(seed=3, dimS=6, dimA=2, lenMin=8, lenMax=40, pTerm=0.5), "after")
    _both(Ls, lambda L: (L.step(3), L.sync()))
    before = [a.copy() for a in Ls[0].get_params()]
    with pytest.raises(capi.HlError):
        Ls[0].step(1)
        Ls[0].sync()
        Ls[0].scalars()
    after = Ls[0].get_params()
    for a, b in zip(before, after):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B,hidden,extra", [(48, (1024, 768), {}), (32, (1000, 520, 640), dict(adv_kind=capi.ADV_GAUSSIAN)),
                                            (2048, (768, 1024), {}), (40, (2048,), dict(adv_kind=capi.ADV_DISCRETE, n_options=5, dimA=1, bounded=[0]))],
                         ids=["1024x768", "1000x520x640-gauss", "b2048-768x1024", "2048-discrete"])
def test_hidden_layers_up_to_2048_units_match_oracle(hip_api, B, hidden, extra):
    """Layer_Base.h:64-113 has no width limit; the library served 512 units until round 5 (the head launch's quarter-of-the-units-per-
    wavefront form stopped there).  Widths up to 2048, not multiples of 16, as last and as inner layers, small and large batches,
    all three heads: per-sample taps and the updated weights against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=9, dimA=3, bounded=[1, 0, 0], hidden=hidden, nnFunc="Tanh", batchSize=B, maxTotObsNum=200000, randSeed=17)
    kw.update(extra)
    G, O = _pair(hip_api, kw, synth_cfg(seed=21, dimS=9, dimA=kw["dimA"], lenMin=20, lenMax=120, pTerm=0.5), 150)
    for _ in range(2):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(3); O.step(3)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    st = np.random.default_rng(5).standard_normal((70, 9)).astype(np.float32)      # rollout inference at these widths (hl_forward)
    assert relinf(G.forward(st[:3]), O.forward(st[:3])) < TOL32 and relinf(G.forward(st), O.forward(st)) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,extra", [((128, 112, 96), dict(adv_kind=capi.ADV_GAUSSIAN, dimA=1, bounded=[1])), ((96, 48), {}),
                                          ((320, 256, 64), dict(adv_kind=capi.ADV_DISCRETE, n_options=4, dimA=1, bounded=[0]))],
                         ids=["128x112x96-gauss", "96x48", "320x256x64-discrete"])
def test_chained_step_equals_the_separate_launches(hip_api, monkeypatch, hidden, extra):
    """Dense nets off the fused kernels: forward chain + head + input-gradient chain as ONE launch (gemm16.hip: step_chain_kernel) against
    the launch list it replaces (SMARTIES_HIP_GENERIC=512: forward chain, head launch, one dX launch per layer) -- the same tile and head
    code in both, so eager and replayed steps must end bit-identical -- and against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=10, dimA=3, bounded=[1, 0, 0], hidden=hidden, nnFunc="Tanh", batchSize=64, maxTotObsNum=20000, randSeed=19)
    kw.update(extra)
    sc = synth_cfg(seed=23, dimS=10, dimA=kw["dimA"], lenMin=5, lenMax=60, pTerm=0.5)
    out = {}
    for tag, env in (("chained", None), ("separate", "512")):
        monkeypatch.delenv("SMARTIES_HIP_GENERIC", raising=False)
        if env:
            monkeypatch.setenv("SMARTIES_HIP_GENERIC", env)
        G, O = _pair(hip_api, kw, sc, 80)
        for _ in range(2):
            G.step(1); O.step(1)
            _compare_step(G, O)
        G.step(37); O.step(37)
        assert relinf(G.get_params()[0], O.get_params()[0]) < 4 * TOL32
        out[tag] = (G.get_params()[0].copy(), G.get_rng_state().copy(), G.scalars().beta, G.scalars().nFarPolicySteps)
        G.close()
    assert np.array_equal(out["chained"][0], out["separate"][0]) and np.array_equal(out["chained"][1], out["separate"][1])
    assert out["chained"][2] == out["separate"][2] and out["chained"][3] == out["separate"][3]


@pytest.mark.gpu
@pytest.mark.parametrize("B,nEps,lenMin,lenMax", [(2048, 900, 60, 160), (4096, 900, 60, 160), (10000, 1500, 40, 100), (16384, 2600, 30, 50)],
                         ids=["b2048", "b4096", "b10000", "b16384-crowded"])
def test_large_batch_sampler_bucket_sort_matches_the_sequential_algorithm(hip_api, B, nEps, lenMin, lenMax):
    """Sample_uniform::sample (Sampling.cpp:82-96) at local batches above 1024 with replays of 70 - 110 thousand transitions -- the range in
    which sample.hip: big_sample_kernel sorts the draws by buckets (below 2^16 transitions it keeps the sorting network) -- from a few
    duplicates per minibatch to a fifth of it (several redraw rounds, merged or sorted again): sorted unique indices, generator state
    and next-state rows against the oracle, minibatch after minibatch."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair
    kw = dict(dimS=4, dimA=2, bounded=[1, 0], hidden=(32, 32), nnFunc="Tanh", batchSize=B, maxTotObsNum=400000, randSeed=29)
    G, O = _pair(hip_api, kw, synth_cfg(seed=31, dimS=4, dimA=2, lenMin=lenMin, lenMax=lenMax, pTerm=0.4), nEps)
    assert G.scalars().nStoredSteps >= 65536
    for _ in range(4):
        G.step(1); O.step(1)
        assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
        assert np.array_equal(G.readback(capi.TAP_STATE), O.readback(capi.TAP_STATE))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden,B,bptt", [(capi.NN_LSTM, (32, 32), 1536, 4), (capi.NN_MGU, (32, 32), 1100, 3), (capi.NN_LSTM, (128, 96), 1280, 3),
                                                (capi.NN_MGU, (96, 80), 2048, 2), (capi.NN_RNN, (24, 16), 1200, 3)],
                         ids=["lstm-2x32", "mgu-2x32", "lstm-128x96", "mgu-96x80-b2048", "rnn-24x16"])
def test_recurrent_nets_at_large_local_batches_match_oracle(hip_api, kind, hidden, B, bptt):
    """Learner_approximator.cpp:67-77 loops over any batch for any network; the library refused recurrent layers above 1024 samples until
    round 5.  The 1024-thread sampler, the per-sample / time-step-major recurrent launches over B windows, the panel head, the weight
    gradients over B x window rows in row chunks: eager and replayed steps against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=6, dimA=2, bounded=[1, 0], hidden=hidden, nnFunc="Tanh", batchSize=B, maxTotObsNum=200000, randSeed=37,
              nn_type=kind, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=bptt)
    G, O = _pair(hip_api, kw, synth_cfg(seed=41, dimS=6, dimA=2, lenMin=2, lenMax=40, pTerm=0.5), 400)
    for _ in range(2):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(5); O.step(5)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_kw,dS", [
    (dict(dimS=576, dimA=2, nAppendedObs=0, conv=[(12, 12, 4, 8, 3, 1), (10, 10, 8, 16, 4, 2)], hidden=(32,), nnFunc="Tanh"), 576),
    (dict(dimS=189, dimA=2, nAppendedObs=0, conv=[(9, 7, 3, 5, 3, 1)], hidden=(32, 24), nnFunc="SoftSign"), 189),
    (dict(dimS=200, dimA=2, bounded=[1, 0], nAppendedObs=3, conv=[(10, 10, 8, 16, 6, 2)], hidden=(40,)), 200),
    (dict(dimS=7056, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=6, nAppendedObs=3, nnFunc="SoftSign", hidden=(64,),
          conv=[(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]), 7056)],
    ids=["two-layers-strided", "odd-geometry", "appended-frames", "racer-atari-stack"])
def test_convolutional_nets_at_large_local_batches_match_oracle(hip_api, cfg_kw, dS):
    """Convolutional preprocessing above 1024 samples per step (refused until round 5): the stacked rows, the per-layer / sample-resident
    convolution launches over 2 x B rows, the filter gradients over B samples, the large-batch sampler and bookkeeping -- against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    atari = dS == 7056      # (the compile-time geometry of RACER_atari.json; the oracle takes ~10 s per step there: fewer of them)
    kw = dict(batchSize=1040 if atari else 1200, maxTotObsNum=60000, randSeed=43); kw.update(cfg_kw)
    G, O = _pair(hip_api, kw, synth_cfg(seed=47, dimS=dS, dimA=kw["dimA"], lenMin=4, lenMax=30, pTerm=0.4), 120 if atari else 250)
    for _ in range(1 if atari else 2):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(2 if atari else 3); O.step(2 if atari else 3)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("algo,B,extra", [("PERrank", 1500, {}), ("PERerr", 2048, {}), ("PERseq", 1280, {}),
                                          ("PERerr", 1100, dict(nn_type=capi.NN_LSTM, nnFunc="Tanh", nnBPTTseq=4))],
                         ids=["PERrank-b1500", "PERerr-b2048", "PERseq-b1280", "PERerr-lstm-b1100"])
def test_prioritised_samplers_at_large_local_batches_follow_the_oracle(hip_api, algo, B, extra):
    """dataSamplingAlgo PERrank / PERerr / PERseq (Sampling.cpp:101-296) above 1024 samples per step (refused until round 5): the
    discrete-distribution draws of the 1024-thread sampler -- two (three) generator words per value in order, duplicates redrawn --
    with the table rebuilt from every step's errors: minibatches, generator state and weights against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=B, maxTotObsNum=100000, randSeed=4, dataSamplingAlgo=algo)
    kw.update(extra)
    G, O = _pair(hip_api, kw, synth_cfg(seed=13, dimS=5, dimA=2, lenMin=20, lenMax=120, pTerm=0.4), 300)
    for k in range(6):
        n = 1 if k % 3 else 2
        G.step(n); O.step(n)
        _compare_step(G, O)
        assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [capi.NN_LSTM, capi.NN_MGU], ids=["lstm", "mgu"])
def test_time_step_major_backward_by_diagonals_is_deterministic(hip_api, kind):
    """rectm.hip: the two producers of a tile of cell deltas run in one launch and whichever arrives second forms the deltas -- from the same
    two inputs either way: two runs must end bit-identical (weights, generator, beta)."""
    from oracle_api import synth_cfg, fill_synth
    sc = synth_cfg(seed=7, dimS=6, dimA=2, lenMin=5, lenMax=60, pTerm=0.4)

    def run():
        L = capi.Learner(hip_api, capi.make_config(dimS=6, dimA=2, bounded=[1, 0], hidden=(128, 96, 64), nnFunc="Tanh", batchSize=72, maxTotObsNum=50000, randSeed=3,
                                                   nn_type=kind, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=9))
        L.init_weights(); fill_synth(L, sc, 200); L.initialize()
        for n in (1, 5, 30):
            L.step(n)
        L.sync()
        out = (L.get_params()[0].copy(), L.get_rng_state().copy(), L.scalars().beta)
        L.close()
        return out
    a, b = run(), run()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


@pytest.mark.gpu
@pytest.mark.parametrize("nOpt,hidden,kind", [(40, (64, 64), capi.NN_FFNN), (64, (96,), capi.NN_FFNN), (48, (160, 96, 64), capi.NN_FFNN), (36, (32, 32), capi.NN_LSTM)],
                         ids=["40-options", "64-options-one-layer", "48-options-chained", "36-options-lstm"])
def test_more_than_32_discrete_options_match_oracle(hip_api, nOpt, hidden, kind):
    """Math/Discrete_policy.h:19-208 and Discrete_advantage have no option limit; the library served 32 (one option per lane of a
    16-lane row, two chunks) until round 5.  Up to 64 options on the head launch (one option per lane of the sample's wavefront):
    per-sample taps, write-backs and the updated weights against the oracle, rollout outputs included."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=8, dimA=1, bounded=[0], hidden=hidden, nnFunc="Tanh", batchSize=48, maxTotObsNum=20000, randSeed=53,
              adv_kind=capi.ADV_DISCRETE, n_options=nOpt, nn_type=kind, nnBPTTseq=4)
    G, O = _pair(hip_api, kw, synth_cfg(seed=57, dimS=8, dimA=1, lenMin=4, lenMax=50, pTerm=0.5), 80)
    for _ in range(2):
        G.step(1); O.step(1)
        _compare_step(G, O)
    G.step(12); O.step(12)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32
    if kind == capi.NN_FFNN:
        st = np.random.default_rng(5).standard_normal((9, 8)).astype(np.float32)
        assert relinf(G.forward(st), O.forward(st)) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("B,hidden,dS", [(512, (64, 64), 9), (1024, (256, 256), 17), (768, (128, 128), 60)], ids=["b512-2x64", "b1024-2x256", "b768-wide-states"])
def test_fused_steps_at_512_to_1024_samples_match_oracle(hip_api, B, hidden, dS):
    """Two-kernel fused steps at 512 - 1024 samples (two to four workgroups per CU, the sampler riders' chains two to four times as
    long as at the bench's 256): eager steps, replayed graphs of several lengths, a discarded pre-drawn minibatch (a given one in
    between): minibatches, generator state and weights against the oracle."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=dS, dimA=3, bounded=[1, 0, 0], hidden=hidden, nnFunc="SoftSign", batchSize=B, maxTotObsNum=200000, randSeed=61)
    G, O = _pair(hip_api, kw, synth_cfg(seed=63, dimS=dS, dimA=3, lenMin=20, lenMax=120, pTerm=0.4), 300)
    for n in (1, 1, 7, 20):
        G.step(n); O.step(n)
        assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
        assert np.array_equal(G.get_rng_state(), O.get_rng_state())
    _compare_step(G, O)
    flat = np.sort(np.random.default_rng(1).choice(G.scalars().nStoredSteps, size=B, replace=False)).astype(np.int64)
    G.step(1, flat=flat); O.step(1, flat=flat)
    _compare_step(G, O)
    G.step(12); O.step(12)
    assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT)) and np.array_equal(G.get_rng_state(), O.get_rng_state())
    assert relinf(G.get_params()[0], O.get_params()[0]) < 2 * TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,B,dS,extra", [((128, 128, 128), 256, 10, dict(adv_kind=capi.ADV_GAUSSIAN, dimA=1, bounded=[1], nnFunc="Tanh")),
                                               ((64, 64, 64), 40, 37, dict(nnFunc="SoftSign")),
                                               ((256, 256, 256), 72, 9, dict(adv_kind=capi.ADV_DISCRETE, n_options=6, dimA=1, bounded=[0], nnFunc="Tanh")),
                                               ((128, 128, 128), 100, 300, dict(dimA=9, bounded=[1, 0, 0, 1, 0, 0, 0, 1, 0], nnFunc="Relu"))],
                         ids=["glider-3x128-gauss", "3x64-vracer", "3x256-discrete", "3x128-wide-state-9-actions"])
def test_three_equal_hidden_blocks_on_the_fused_kernel_match_oracle(hip_api, monkeypatch, hidden, B, dS, extra):
    """fusedw.hip with a third hidden block (settings/RACER_glider.json is 3 x 128 under the Gaussian advantage): forward, head and
    input gradients of the whole minibatch in one launch -- four panel barriers instead of two --, then the weight gradients.  Per-sample
    taps, write-backs, weights (eager and replayed steps, next-state rows of truncated episodes) against the oracle; the separate
    launches (SMARTIES_HIP_GENERIC=1) must land within the same tolerance of it."""
    from oracle_api import synth_cfg
    from test_hip_parity import _pair, _compare_step
    kw = dict(dimS=dS, dimA=3, bounded=[1, 0, 0], hidden=hidden, batchSize=B, maxTotObsNum=50000, randSeed=71)
    kw.update(extra)
    sc = synth_cfg(seed=73, dimS=dS, dimA=kw["dimA"], lenMin=3, lenMax=50, pTerm=0.5)
    for env in (None, "1"):
        monkeypatch.delenv("SMARTIES_HIP_GENERIC", raising=False)
        if env:
            monkeypatch.setenv("SMARTIES_HIP_GENERIC", env)
        G, O = _pair(hip_api, kw, sc, 120)
        for _ in range(3):
            G.step(1); O.step(1)
            _compare_step(G, O)
        G.step(25); O.step(25)
        assert np.array_equal(G.readback(capi.TAP_FLAT), O.readback(capi.TAP_FLAT))
        assert relinf(G.get_params()[0], O.get_params()[0]) < 4 * TOL32      # (28 steps of two fp32 summation orders: the weights, not the last step's taps)
        G.close()
