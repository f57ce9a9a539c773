"""Step time at cfg-NS with truncated episodes (as bench.py: ~1.3 next-state rows per minibatch -> one more panel in K1) against
terminated ones (no next-state rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
for term in (0, 1, 0, 1):
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(bench.N_EPISODES):
        ep = bench.synthetic_episode(np, e); ep["terminated"] = term
        L.append_episode(**ep)
    L.initialize(); L.step(2000); L.sync()
    t0 = time.perf_counter(); L.step(8000); L.sync(); dt = time.perf_counter() - t0
    print("terminated=%d: %.2f us per step; K1 %.2f (no rider %.2f), K2 %.2f" % (term, dt / 8000 * 1e6, L.kernel_profile(26, 200), L.kernel_profile(28, 200), L.kernel_profile(27, 200)))
    L.close()
