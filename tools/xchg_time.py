"""Two replicas of the cfg-NS learner (batch 128 each) in ONE process on one GPU, exchanging through each other's windows
(xchg.hip): microseconds per step and, under rocprofv3 --kernel-trace --stats, the exchange kernel's duration (its minimum is the
push + sum + Adam + bookkeeping time when the peer's message is already there; both replicas share the device here)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
Ls = []
for r in range(2):
    L = capi.Learner(api, capi.make_config(n_ranks=2, rank=r, **bench.CFG)); L.init_weights()
    for e in range(r * 2500, (r + 1) * 2500):
        L.append_episode(**bench.synthetic_episode(np, e))
    Ls.append(L)
hs = [L.xchg_export() for L in Ls]
def both(fn):
    ts = [threading.Thread(target=fn, args=(L,)) for L in Ls]
    [t.start() for t in ts]; [t.join() for t in ts]
both(lambda L: (L.xchg_connect(hs), L.initialize(), L.step(200), L.sync()))
t0 = time.perf_counter(); both(lambda L: (L.step(4000), L.sync())); dt = time.perf_counter() - t0
print("two replicas on one device: %.1f us per step (both), weights identical: %s" % (dt / 4000 * 1e6, np.array_equal(Ls[0].get_params()[0], Ls[1].get_params()[0])))
