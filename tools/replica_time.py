"""NR replicas of cfg-NS in ONE process on one device (a thread each): wall time per replayed step, folded step against the round-5
three-launch step (default; the folded one: SMARTIES_HIP_FOLD=1) and the un-pushed one.  One device shared by all replicas: an upper bound of a node's step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "30000")
import numpy as np
import torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
import test_hip_r6 as t6
api = load_hip()
nr = int(os.environ.get("NR", "2"))
B = int(os.environ.get("BATCH", "256"))
cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=B, maxTotObsNum=262144, randSeed=42)
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=100, lenMax=200, pTerm=0.3)
X = t6._replicas(api, cfg_kw, sc, nr, 60 * nr, True)
t6._both(X, lambda L: (L.step(64), L.sync()))
for n in (64, 512, 512):
    t0 = time.perf_counter()
    t6._both(X, lambda L: (L.step(n), L.sync()))
    dt = time.perf_counter() - t0
print("replicas %d  local batch %d  FOLD=%s NO_PUSH=%s : %.2f us per step (512 replayed steps, all replicas on one device)" % (
    nr, X[0].B, os.environ.get("SMARTIES_HIP_FOLD", "0"), os.environ.get("SMARTIES_HIP_NO_PUSH", "0"), dt / n * 1e6))
