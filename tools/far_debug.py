"""Far-policy count (ReplayStats::nFarPolicySteps) of device and oracle after each step of a reference fixture, with the
per-episode fractions both hold and the reference's float-add/truncate loop run on the host over either set of fractions.
usage: far_debug.py <fixture.bin>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip
import parity
from oracle_api import oracle_api

name = sys.argv[1]
fx = parity.load_fixture(name)
api = load_hip(); orc = oracle_api()
L = capi.Learner(api, parity.fixture_config(fx)); parity.setup_from_fixture(L, fx)
O = capi.Learner(orc, parity.fixture_config(fx)); parity.setup_from_fixture(O, fx)


def seq(N, F):
    n = 0
    for a, f in zip(N, F):
        n = int(np.float32(np.float32(n) + np.float32(a) * np.float32(f)))
    return n


nEp = int(fx["cfg"][3])
for k in range(1, int(fx["cfg"][4]) + 1):
    sk = "s%d_" % k
    if sk + "flat" not in fx:
        break
    flat = parity.flat_for(L, fx[sk + "tag"], fx[sk + "t"]); order = np.argsort(flat, kind="stable")
    L.step(1, flat=flat[order]); O.step(1, flat=flat[order])
    N = [L.episode_info(p)[1] for p in range(nEp)]
    fd = np.array([L.episode_stats(p)[2] for p in range(nEp)]); fo = np.array([O.episode_stats(p)[2] for p in range(nEp)])
    print(k, "device", L.scalars().nFarPolicySteps, "oracle", O.scalars().nFarPolicySteps, "fixture", fx["traj_nfar"][k - 1],
          "| host loop over device fractions", seq(N, fd), "over oracle fractions", seq(N, fo), "| fractions differ at", np.nonzero(fd != fo)[0][:8],
          "Cmax", L.scalars().CmaxRet)
