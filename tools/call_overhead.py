"""Wall time of hl_step(n) + hl_sync for small n at cfg-NS (what a caller that steps a few gradient steps at a time --
and the round driver's `bench.py --steps 20` -- sees): fixed cost per call vs per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
import bench
from smarties_amd import capi, load_hip

api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG))
L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
L.step(300); L.sync()
res = {}
for n in (1, 2, 4, 8, 16, 20, 32, 64, 128, 256):
    reps = max(20, 2000 // n)
    L.step(n); L.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        L.step(n); L.sync()
    dt = (time.perf_counter() - t0) / reps
    res[n] = dt
    print("n = %3d: %8.1f us per call, %6.2f us per step" % (n, dt * 1e6, dt * 1e6 / n))
# back-to-back calls without a sync in between (a host loop that only enqueues)
for n in (1, 4, 20):
    reps = 4000 // n
    L.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        L.step(n)
    L.sync(); dt = (time.perf_counter() - t0) / reps
    print("n = %3d enqueue only: %8.1f us per call, %6.2f us per step" % (n, dt * 1e6, dt * 1e6 / n))

# the round driver's protocol: barrier + torch.cuda.synchronize() on both sides of the timed region
def barrier():
    torch.cuda.synchronize(); L.sync()
for n in (20, 200):
    ts = []
    for _ in range(30):
        barrier(); t0 = time.perf_counter(); L.step(n); L.sync(); barrier(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("bench protocol n = %3d: median %.1f us (%.2f per step), min %.1f" % (n, ts[15] * 1e6, ts[15] * 1e6 / n, ts[0] * 1e6))
ts = []
for _ in range(50):
    L.sync(); t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); print("torch.cuda.synchronize() on an idle device: median %.1f us" % (ts[25] * 1e6))
ts = []
for _ in range(50):
    t0 = time.perf_counter(); L.sync(); ts.append(time.perf_counter() - t0)
ts.sort(); print("hl_sync on an idle stream: median %.1f us" % (ts[25] * 1e6))
