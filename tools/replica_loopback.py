"""ONE replica of an N-replica job stepping alone on the device ("loopback"): its peers' windows exist (idle twin learners) and their
arrival stamps in ITS window are preset to a huge sequence number, so it never waits -- it pushes its gradient into N - 1 windows,
sums N slots (its own + the twins' stale ones), applies Adam and closes the step.  What a replica's step costs when nobody competes
for its CUs and every peer is already there: the floor of a node's step (plus the links).  The numbers mean nothing numerically.
  python tools/replica_loopback.py            (env NR=2|4|8, BATCH=global batch, SMARTIES_HIP_FOLD=1 / SMARTIES_HIP_NO_PUSH=1)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "20000")
import numpy as np
import torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
import test_hip_r6 as t6
hip = C.CDLL("libamdhip64.so")
api = load_hip()
nr = int(os.environ.get("NR", "2"))
B = int(os.environ.get("BATCH", "256"))
steps = int(os.environ.get("STEPS", "512"))
cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=B, maxTotObsNum=262144, randSeed=42)
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=100, lenMax=200, pTerm=0.3)
X = t6._replicas(api, cfg_kw, sc, nr, 60 * nr, True)
hd = bytes(X[0].xchg_export())
addr = int.from_bytes(hd[64:72], "little")
flags = np.zeros((2, nr, 64), np.uint64)
assert hip.hipMemcpy(C.c_void_p(flags.ctypes.data), C.c_void_p(addr), C.c_size_t(flags.nbytes), 2) == 0
flags[:, 1:, :] = np.uint64(1) << np.uint64(62)
assert hip.hipMemcpy(C.c_void_p(addr), C.c_void_p(flags.ctypes.data), C.c_size_t(flags.nbytes), 1) == 0
L = X[0]
L.step(64); L.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); L.step(steps); L.sync(); best = min(best, time.perf_counter() - t0)
print("loopback: replica 0 of %d, local batch %d, FOLD=%s NO_PUSH=%s : %.2f us per step (%d replayed steps)" % (
    nr, L.B, os.environ.get("SMARTIES_HIP_FOLD", "0"), os.environ.get("SMARTIES_HIP_NO_PUSH", "0"), best / steps * 1e6, steps), flush=True)
os._exit(0)      # (the twins never stepped: nothing to wait for)
