"""hl_kernel_profile of the two step kernels with (26, 27) and without (28, 29) their rider workgroups, cfg-NS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.step(2000); L.sync()
for rep in range(2):
    print({pid: round(L.kernel_profile(pid, 200), 2) for pid in (12, 26, 28, 27, 29, 7)})
