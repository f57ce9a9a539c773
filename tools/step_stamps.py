"""Where the time of a short call goes (the round driver's protocol: hl_step(20) + hl_sync on an idle device): entry stamps of the
two step kernels for every step of the call (library built with HL_EXTRA_FLAGS=-DHL_STEP_STAMPS), the host's wall time around the
call next to the device's span.  usage: step_stamps.py [steps per call] [calls]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
import bench
from smarties_amd import capi, load_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
api = load_hip()
g = api.lib.hl_debug_step_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
if not os.environ.get("NO_PREPARE"):
    L.prepare_steps(5); L.prepare_steps(n)
L.step(5); L.sync()
rows = []
for c in range(calls):
    time.sleep(0.002)
    g0 = L.scalars().nGradSteps
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns(); L.step(n); ts = time.perf_counter_ns(); L.sync(); t1s = time.perf_counter_ns(); torch.cuda.synchronize(); t1 = time.perf_counter_ns()
    out = (C.c_longlong * 128)(); assert g(L.h, out) == 0
    st = np.array(list(out), dtype=np.int64)
    k1 = np.array([st[(g0 + j) & 63] for j in range(n)]) * 10              # ns (100 MHz clock)
    k2 = st[64:128] * 10
    k2 = np.sort(k2[(k2 >= k1[0]) & (k2 <= k1[-1] + 30000)])[:n]
    rows.append((t1 - t0, k1, k2, ts - t0, t1s - t0))
    per = np.diff(k1)
    print("call %2d: host %7.1f us = %.2f us/step | device first K1 -> last K1 %.1f us | K1->K1: first %.2f  median %.2f  max %.2f (at %d) | K1->K2 median %.2f"
          % (c, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / n, (k1[-1] - k1[0]) / 1e3, per[0] / 1e3, np.median(per) / 1e3, per.max() / 1e3, per.argmax(),
             np.median(k2[:len(k1)] - k1[:len(k2)]) / 1e3 if len(k2) else -1))
host = np.median([r[0] for r in rows[2:]]) / 1e3
span = np.median([r[1][-1] - r[1][0] for r in rows[2:]]) / 1e3
print("median host times (us): hl_step returned %.1f, hl_sync returned %.1f, torch.cuda.synchronize returned %.1f" % (
    np.median([r[3] for r in rows[2:]]) / 1e3, np.median([r[4] for r in rows[2:]]) / 1e3, host))
print("median: host %.1f us per call, device span of %d K1 entries %.1f us -> per step inside the call %.2f us, around it %.1f us"
      % (host, n, span, span / (n - 1), host - span / (n - 1) * n))
allper = np.array([np.diff(r[1]) for r in rows[2:]]) / 1e3
print("per-step interval by position in the call (median over calls):", " ".join("%.1f" % v for v in np.median(allper, axis=0)))
