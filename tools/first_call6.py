"""The round driver's protocol (`bench.py --steps 20 --warmup 5`) on a fresh learner as bench.py runs it now (graphs of 5 and 20
steps prepared, nothing but the warm-up between set-up and the timed call), with a few milliseconds of unrelated device work
put in front of the timed region -- does the first timed call run at working clocks then?  variants by argv[1]:
  none       as bench.py
  mm-before  torch matmuls (about 5 ms) before the warm-up steps
  mm-between torch matmuls between the warm-up steps and the timed call
  steps      200 extra learner steps (one call) before the warm-up (what more warm-up would do)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, bench
from smarties_amd import capi, load_hip
api = load_hip()
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
def barrier():
    L.sync(); torch.cuda.synchronize()
A = torch.randn(4096, 4096, device="cuda"); Bm = torch.randn(4096, 4096, device="cuda")
def load(ms=5.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        torch.mm(A, Bm); torch.cuda.synchronize()
load(0.1); barrier()
L.prepare_steps(5); L.prepare_steps(20)
if mode == "steps":
    L.step(200); barrier()
if mode == "mm-before":
    load()
L.step(5); barrier()
if mode == "mm-between":
    load()
t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
out = [dt * 1e6]
for i in range(3):
    t0 = time.perf_counter(); L.step(20); barrier(); out.append((time.perf_counter() - t0) * 1e6)
print("%-10s first timed call %.1f us (%.2f per step); again: %s" % (mode, out[0], out[0] / 20, " ".join("%.1f" % v for v in out[1:])))
