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
    sample-resident kernel, (TAIL=2) the any-geometry kernel on every stack, (TAIL=0) the per-layer launches of conv.hip; filter
    gradients staged in LDS (default) or gathered (DW_G=0).  Each follows the reference's taps; among themselves they differ by
    summation order only."""
    ws = {}
    for tag, env in (("default", {}), ("any_geometry", {"SMARTIES_HIP_CONV_TAIL": "2"}), ("per_layer", {"SMARTIES_HIP_CONV_TAIL": "0"}), ("backward_only", {"SMARTIES_HIP_CONV_TAIL": "3"}),
                     ("gather_dw", {"SMARTIES_HIP_CONV_DW_G": "0"}), ("one_row_dw", {"SMARTIES_HIP_CONV_DW_G": "1"})):
        for k in ("SMARTIES_HIP_CONV_TAIL", "SMARTIES_HIP_CONV_DW_G"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ws[tag] = run_fixture(hip_api, name)
    ref = ws["default"]
    for tag, w in ws.items():
        assert np.abs(w - ref).max() <= 1e-5 * np.abs(ref).max(), tag
