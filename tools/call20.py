"""the driver's window (5 warm-up steps, then ONE timed call of 20 steps bracketed by torch.cuda.synchronize + hl_sync) taken apart:
host time of hl_step(20), of the device synchronisation and of hl_sync, for the first and the following calls (python tools/call20.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from smarties_amd import capi, load_hip
api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
L.prepare_steps(5); L.prepare_steps(20)
if os.environ.get("IDLE_MS"):
    time.sleep(float(os.environ["IDLE_MS"]) / 1e3)
if os.environ.get("SPIN_MS"):      # keep the device busy (clocks, power state) right up to the warm-up steps
    x = torch.ones(1 << 22, device="cuda"); t_end = time.perf_counter() + float(os.environ["SPIN_MS"]) / 1e3
    while time.perf_counter() < t_end:
        for _ in range(50):
            x.mul_(1.0000001)
    torch.cuda.synchronize()
if os.environ.get("SPIN_OWN"):      # ... with the learner's own kernels: rollout inference of one state (changes nothing of the learner)
    st = np.zeros((1, 17), np.float32); t_end = time.perf_counter() + float(os.environ["SPIN_OWN"]) / 1e3
    while time.perf_counter() < t_end:
        L.forward(st)
L.step(5); torch.cuda.synchronize(); L.sync()
for i in range(6):
    t0 = time.perf_counter(); L.step(20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter(); L.sync(); t3 = time.perf_counter()
    print("call %d: total %.1f us = %.2f us/step | hl_step returns after %.1f, device synchronised after +%.1f, hl_sync +%.1f" % (
        i + 1, (t3 - t0) * 1e6, (t3 - t0) * 1e6 / 20, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6), flush=True)
