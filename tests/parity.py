"""Shared parity machinery: drive a learner (HIP library or CPU oracle -- same C-ABI) through a
golden fixture recorded from the compiled reference and report the deviations."""
import os

import numpy as np

from golden_io import load_blob
from oracle_api import fill_synth, synth_cfg
from smarties_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return load_blob(os.path.join(GOLDEN, name))


def fixture_config(fx, **over):
    dS, dA, B = [int(x) for x in fx["cfg"][:3]]
    hp = fx["hp"]
    func = over.pop("nnFunc", None)
    kw = dict(dimS=dS, dimA=dA, bounded=fx["bounded"], hidden=[int(x) for x in fx["layers"]], batchSize=B,
              maxTotObsNum=int(hp[12]), clipImpWeight=hp[0], penalTol=hp[1], epsAnneal=hp[2], gamma=hp[3],
              lambda_=hp[4], learnrate=hp[5], explNoise=hp[6], outWeightsPrefac=hp[7], nnLambda=hp[8],
              randSeed=42, nnFunc=func or "SoftSign")
    kw.update(over)
    return capi.make_config(**kw)


def fixture_synth(fx):
    dS, dA = int(fx["cfg"][0]), int(fx["cfg"][1])
    hp = fx["hp"]
    return synth_cfg(seed=int(fx["cfg"][8]), dimS=dS, dimA=dA, lenMin=int(fx["cfg"][9]), lenMax=int(fx["cfg"][10]),
                     pTerm=hp[9], muSpread=hp[10], actNoise=hp[11])


def relinf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def setup_from_fixture(L, fx, use_fixture_weights=False):
    """create -> init weights -> fill synthetic replay -> initializeLearner."""
    nEps = int(fx["cfg"][3])
    if use_fixture_weights:
        L.set_params(w=fx["W0"])
        L.set_rng_state(fx["rng_before_init"])
    else:
        L.init_weights()
    fill_synth(L, fixture_synth(fx), nEps)
    L.initialize()


def tags_of(L):
    n = L.scalars().nStoredEps
    return np.array([L.episode_info(k)[0] for k in range(n)], np.int64)


def episode_arrays_by_tag(L, field):
    out = {}
    for k in range(L.scalars().nStoredEps):
        out[L.episode_info(k)[0]] = L.episode_field(k, field)
    return out


def fixture_arrays_by_tag(fx, tags_key, arr_key, lens):
    out, o = {}, 0
    for tag in fx[tags_key]:
        n = lens[int(tag)]
        out[int(tag)] = fx[arr_key][o:o + n]
        o += n
    return out
