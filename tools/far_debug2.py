"""Far-policy count of device and oracle drawing their own minibatches (tests/test_hip_parity.py: _pair), with the fractions
both hold and the reference's loop run on the host over either set.  usage: far_debug2.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
from parity import far_count_loop, storage_order
import test_hip_parity as T

cfg_kw = dict(dimS=257, dimA=17, bounded=[0] * 17, hidden=(256, 256), batchSize=32, maxTotObsNum=60000, clipImpWeight=(17 / 2.0) ** 0.5, randSeed=33)
sc_kw = dict(seed=29, dimS=257, dimA=17, lenMin=20, lenMax=120, pTerm=0.5, muSpread=0.2)
G, O = T._pair(load_hip(), cfg_kw, synth_cfg(**sc_kw), 40)
for k in range(8):
    G.step(1); O.step(1)
    n = G.scalars().nStoredEps
    fd = np.array([G.episode_stats(p)[2] for p in range(n)]); fo = np.array([O.episode_stats(p)[2] for p in range(n)])
    N = np.array([G.episode_info(p)[1] for p in range(n)])
    print(k, "device", G.scalars().nFarPolicySteps, "oracle", O.scalars().nFarPolicySteps, "| host loop over device fractions",
          far_count_loop(G, storage_order(G)), "over oracle fractions", far_count_loop(O, storage_order(O)),
          "| fractions differ at", np.nonzero(fd != fo)[0][:8], "same order", storage_order(G) == storage_order(O))
    if k == 7:
        print("terms", (N.astype(np.float32) * fd.astype(np.float32)).tolist())
