"""Is the extra time of a first timed call a property of the graph executable (its first execution) or of the device state?
Graphs of 20, 21 and 22 steps are prepared; after the 20-step call has been repeated until it is fast, the 21-step graph runs for
the first time, then again; then 22 after 50 ms of idling."""
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
    L.sync(); torch.cuda.synchronize()
for n in (5, 20, 21, 22):
    L.prepare_steps(n)
L.step(5); barrier()
def call(n, label):
    t0 = time.perf_counter(); L.step(n); barrier(); dt = (time.perf_counter() - t0) * 1e6
    print("%-34s %6.1f us  (%.2f per step)" % (label, dt, dt / n))
call(20, "20 steps, first execution"); call(20, "20 steps again"); call(20, "20 steps again"); call(20, "20 steps again")
call(21, "21 steps, first execution"); call(21, "21 steps again"); call(21, "21 steps again")
call(20, "20 steps again")
time.sleep(0.05)
call(22, "22 steps, first, after 50 ms idle"); call(22, "22 steps again"); call(22, "22 steps again")
