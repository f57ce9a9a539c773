"""Latency of hl_forward (RACER::selectAction's network outputs for raw states, Learners/RACER.cpp:30-59): what an env-service
thread pays per agent step, at cfg-NS and for a batch of agents."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(200):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.step(10); L.sync()
g = np.random.default_rng(0)
for n in (8, 1, 2, 4, 8, 1, 64, 256):
    st = g.standard_normal((n, 17)).astype(np.float32)
    for _ in range(20): L.forward(st)
    t0 = time.perf_counter()
    for _ in range(500): L.forward(st)
    dt = (time.perf_counter() - t0) / 500
    print("hl_forward of %3d states: %.1f us per call" % (n, dt * 1e6))
# while the learner trains (steps enqueued on the same stream)
st = g.standard_normal((1, 17)).astype(np.float32)
L.step(2000)
t0 = time.perf_counter(); L.forward(st); dt = time.perf_counter() - t0
print("hl_forward behind 2000 queued steps: %.1f us" % (dt * 1e6))
