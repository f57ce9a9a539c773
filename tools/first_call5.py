"""The driver's protocol with different things in front of the warm-up steps (one process per variant):
  cold      nothing: seconds of host-only set-up, then 5 warm-up steps, barrier, 20 timed steps
  other     kernel-profile passes of a SECOND learner (its own stream) first -- what bench.py did in round 2
  own       empty-kernel profile passes on the measured learner's OWN stream first (no state of the learner changes)
  own+k     ... plus profile passes of its own step kernels afterwards?  no: those advance the learner -- not a variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, bench
from smarties_amd import capi, load_hip
api = load_hip()
mode = sys.argv[1]
def make():
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(bench.N_EPISODES):
        L.append_episode(**bench.synthetic_episode(np, e))
    L.initialize(); L.prepare_steps(5); L.prepare_steps(20)
    return L
L = make()
def barrier():
    L.sync(); torch.cuda.synchronize()
if mode == "other":
    P = make(); P.step(64)
    for pid in (12, 26, 27):
        P.kernel_profile(pid, 200)
if mode.startswith("own"):
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
        L.kernel_profile(12, 400)
L.step(5); barrier()
t0 = time.perf_counter(); L.step(20); barrier(); dt = time.perf_counter() - t0
out = ["%.1f" % (dt * 1e6)]
for i in range(4):
    t0 = time.perf_counter(); L.step(20); barrier(); out.append("%.1f" % ((time.perf_counter() - t0) * 1e6))
print(mode, " ".join(sys.argv[2:]), "timed call, then again x4:", " ".join(out))
