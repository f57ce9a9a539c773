"""Is the first-call surcharge of the driver's protocol a property of the learner (its pages, its graphs) or of the device
(clocks)?  Learner L is warmed with repeated 20-step calls; then a second learner Q, prepared the same way but never stepped,
runs the protocol (5 warm-up steps, barrier, 20 timed steps) right behind L's warm calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, bench
from smarties_amd import capi, load_hip
api = load_hip()
def make():
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(bench.N_EPISODES):
        L.append_episode(**bench.synthetic_episode(np, e))
    L.initialize(); L.prepare_steps(5); L.prepare_steps(20)
    return L
L, Q = make(), make()
def timed(X, n=20):
    t0 = time.perf_counter(); X.step(n); X.sync(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
L.step(5); L.sync()
print("L:", " ".join("%.1f" % timed(L) for _ in range(6)))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "touch":
    Q.prepare_steps(20)
Q.step(5); Q.sync(); torch.cuda.synchronize()
print("Q first:", "%.1f" % timed(Q), "then", " ".join("%.1f" % timed(Q) for _ in range(4)))
print("L again:", " ".join("%.1f" % timed(L) for _ in range(3)))
# how many steps until a fresh learner runs at the sustained rate: per-call times of 5-step calls from the start
R = make()
R.prepare_steps(20)
print("R 5-step calls from its first step:", " ".join("%.1f" % timed(R, 5) for _ in range(12)))
