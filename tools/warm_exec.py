"""Is a whole-call graph executable fast at its next launch if NO other executable was launched in between (short calls issued directly)?
usage: SMARTIES_HIP_EAGER_CHAIN=8 python tools/warm_exec.py   -- prints us per step of 20-step calls in several sequences"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
api = load_hip()
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=256, maxTotObsNum=1000000))
L.init_weights(); fill_synth(L, sc, 5000); L.initialize()
def bar():
    torch.cuda.synchronize(); L.sync()
def timed(n):
    bar(); t0 = time.perf_counter(); L.step(n); bar(); return (time.perf_counter() - t0) / n * 1e6
if os.environ.get("NO_PREPARE") != "1":      # (NO_PREPARE=1 SMARTIES_HIP_EAGER_CHAIN=64: every call below is issued as direct launches)
    L.prepare_steps(20)
print("first launch of the 20-step executable          %.2f us/step" % timed(20))
print("again                                           %.2f" % timed(20))
for k in (1, 3, 5, 8):
    L.step(k); print("after a %d-step call                             %.2f" % (k, timed(20)))
time.sleep(0.01); print("after 10 ms of idling                            %.2f" % timed(20))
L.step(5); time.sleep(0.002); print("after a 5-step call and 2 ms                     %.2f" % timed(20))
if os.environ.get("NO_PREPARE") != "1":
    L.prepare_steps(5)
L.step(5); print("after a 5-step call through ITS executable       %.2f" % timed(20))
print("again                                           %.2f" % timed(20))
