"""The round driver's protocol in a loop: a W-step call, barrier, a K-step call, barrier -- where does the first K-step call behind a
call of another size lose its ~17 us (launch, every step, or the end)?  Entry stamps of the step kernels (library built with
HL_EXTRA_FLAGS=-DHL_STEP_STAMPS) next to the host's clock.  usage: alt_calls.py [W] [K] [rounds]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from smarties_amd import capi, load_hip

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
api = load_hip()
g = api.lib.hl_debug_step_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
if not os.environ.get("NO_PREPARE"):
    L.prepare_steps(W); L.prepare_steps(K)


def barrier():
    torch.cuda.synchronize(); L.sync()


def call(n):
    g0 = L.scalars().nGradSteps
    barrier()
    t0 = time.perf_counter_ns(); L.step(n); barrier(); t1 = time.perf_counter_ns()
    out = (C.c_longlong * 128)(); assert g(L.h, out) == 0
    st = np.array(list(out), dtype=np.int64)
    k1 = np.array([st[(g0 + j) & 63] for j in range(n)]) * 10
    return (t1 - t0) / 1e3, k1


for r in range(rounds):
    tw, _ = call(W)
    t1, k1 = call(K)
    t2, k2 = call(K)
    d1, d2 = np.diff(k1) / 1e3, np.diff(k2) / 1e3
    print("round %d: W-call %.1f us | first K-call %.1f us (device span %.1f, steps: first %.2f median %.2f max %.2f) | same call again %.1f us (device span %.1f, median step %.2f)"
          % (r, tw, t1, (k1[-1] - k1[0]) / 1e3, d1[0], np.median(d1), d1.max(), t2, (k2[-1] - k2[0]) / 1e3, np.median(d2)))
