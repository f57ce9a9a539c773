"""What the round driver's barrier costs after a call: hl_sync (polled completion word) followed by torch.cuda.synchronize().
usage: sync_cost.py   (environment variables of the HIP runtime are taken from the caller)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from smarties_amd import capi, load_hip

api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.prepare_steps(20); L.step(20); L.sync(); torch.cuda.synchronize()


def med(f, n=30):
    v = []
    for _ in range(n):
        time.sleep(0.001)
        v.append(f())
    return np.median(v) / 1e3


def idle_sync():
    t0 = time.perf_counter_ns(); torch.cuda.synchronize(); return time.perf_counter_ns() - t0


def call(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns(); L.step(20)
    if kind == "poll+dev":
        L.sync(); t1 = time.perf_counter_ns(); torch.cuda.synchronize()
    elif kind == "dev":
        t1 = t0; torch.cuda.synchronize()
    elif kind == "dev+poll":
        torch.cuda.synchronize(); t1 = time.perf_counter_ns(); L.sync()
    else:
        L.sync(); t1 = time.perf_counter_ns()
    t2 = time.perf_counter_ns()
    return (t2 - t0, t1 - t0)


print("torch.cuda.synchronize() on an idle device: %.1f us" % med(idle_sync))
for kind in ("poll", "poll+dev", "dev", "dev+poll"):
    r = [call(kind) for _ in range(30)][5:]
    print("%-9s: call of 20 steps %.1f us (hl_sync returned at %.1f)" % (kind, np.median([a for a, _ in r]) / 1e3, np.median([b for _, b in r]) / 1e3))
