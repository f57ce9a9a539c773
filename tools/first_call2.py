"""first_call.py with the order bench.py could use: the roofline passes (hl_kernel_profile) first, then warm-up and the timed call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, bench
from smarties_amd import capi, load_hip
api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
def barrier():
    torch.cuda.synchronize(); L.sync()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "profile":
    for pid in (12, 26, 27, 28):
        try: L.kernel_profile(pid, 200)
        except Exception as e: print("pid", pid, e)
if mode.startswith("k"):
    for pid in mode[1:].split(","):
        L.kernel_profile(int(pid), 200)
if mode == "sleep":
    time.sleep(2.0)
L.step(5); barrier()
t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
print(mode, "first timed call: %.1f us (%.2f per step)" % (dt * 1e6, dt * 1e6 / 20))
for i in range(3):
    t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
    print("   again: %.1f us (%.2f per step)" % (dt * 1e6, dt * 1e6 / 20))
