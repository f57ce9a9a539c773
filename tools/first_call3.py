"""The round driver's protocol (`bench.py --steps 20 --warmup 5`) on a fresh learner, variants by argv[1]:
  stock      the stock graph sizes (16 + 4), hipStreamSynchronize
  prep       hl_prepare_steps(5), hl_prepare_steps(20): one graph per call, completion stamp polled by hl_sync
  prep+run   as prep, and the 20-step graph has been launched once before (20 extra steps right after initialize)
Each preceded by kernel-profile passes on a second learner, like bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, bench
from smarties_amd import capi, load_hip
api = load_hip()
mode = sys.argv[1] if len(sys.argv) > 1 else "stock"
def make():
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(bench.N_EPISODES):
        L.append_episode(**bench.synthetic_episode(np, e))
    L.initialize()
    return L
L = make()
def barrier():
    L.sync(); torch.cuda.synchronize()
if mode.startswith("prep"):
    L.prepare_steps(5); L.prepare_steps(20)
if mode == "prep+run":
    L.step(20); barrier()
P = make(); P.step(64)
for pid in (12, 26, 27):
    P.kernel_profile(pid, 200)
if mode in ("touch", "touch+run"):     # (library built with the touch pass: SMARTIES_HIP_TOUCH=1 makes hl_prepare_steps read one word per 4 KB of the replay)
    L.prepare_steps(5); L.prepare_steps(20)
if mode == "late":      # the measured learner is built AFTER the profile passes
    L.close(); L = make(); L.prepare_steps(5); L.prepare_steps(20)
if mode == "touch+run":
    L.step(20); barrier()
L.step(5); barrier()
t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
print(mode, "first timed call: %.1f us (%.2f per step)" % (dt * 1e6, dt * 1e6 / 20))
for i in range(4):
    t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
    print("   again: %.1f us (%.2f per step)" % (dt * 1e6, dt * 1e6 / 20))
for d in (0.0002, 0.001, 0.01, 0.1, 1.0):     # the same call after the device has idled for d seconds
    ts = []
    for _ in range(3):
        time.sleep(d); t0 = time.perf_counter(); L.step(20); barrier(); ts.append((time.perf_counter() - t0) * 1e6)
    print("   after %.4f s idle: %s us" % (d, " ".join("%.1f" % t for t in ts)))
barrier(); t0 = time.perf_counter(); L.step(900); barrier(); dt = time.perf_counter() - t0
print("   900 steps: %.2f us per step" % (dt * 1e6 / 900))
