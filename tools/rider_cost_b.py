import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
api = load_hip()
for B in (256, 512, 1024):
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
    L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=B, maxTotObsNum=1048576))
    L.init_weights(); fill_synth(L, sc, 2500); L.initialize(); L.step(200); L.sync()
    print(B, {pid: round(L.kernel_profile(pid, 200), 2) for pid in (26, 28, 27, 29)}, flush=True)
    L.close()
