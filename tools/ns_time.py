"""cfg-NS (BASELINE.json's metric configuration) on one GPU: us per replayed step, best of REPS calls of STEPS steps (python tools/ns_time.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from smarties_amd import capi, load_hip
api = load_hip()
steps, reps = int(os.environ.get("STEPS", "2000")), int(os.environ.get("REPS", "6"))
CFG = dict(bench.CFG)
if os.environ.get("BATCH"):
    CFG["batchSize"] = int(os.environ["BATCH"])
L = capi.Learner(api, capi.make_config(**CFG)); L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
L.step(300); L.sync()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); L.step(steps); L.sync(); ts.append((time.perf_counter() - t0) / steps * 1e6)
print("%s: batch %d: %.2f us per step (best of %d calls of %d steps; all: %s)" % (os.environ.get("TAG", "run"), CFG["batchSize"], min(ts), reps, steps, " ".join("%.2f" % t for t in ts)), flush=True)
