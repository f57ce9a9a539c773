"""(tm_stamps.py: the same run with the device time stamps of rectm.hip: lstm_tm_fwd_kernel; library built with HL_EXTRA_FLAGS=-DHL_TM_STAMPS)
LSTM (or KIND=mgu: MGU) layers wider than 64 cells (rectm.hip): us per replayed step of 2 x N cells, batch 128, BPTT 16; under rocprofv3 its kernels.  usage: lstm_wide_time.py [cells] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip
api = load_hip()
nC = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kw = dict(dimS=4, dimA=1, bounded=[1], hidden=(nC, nC), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144, gamma=0.99, adv_kind=capi.ADV_GAUSSIAN,
          nn_type={"lstm": capi.NN_LSTM, "mgu": capi.NN_MGU}[os.environ.get("KIND", "lstm")], nnLambda=1e-6, explNoise=0.1)
g = np.random.default_rng(5)
L = capi.Learner(api, capi.make_config(randSeed=7, **kw)); L.init_weights()
for e in range(300):
    N = 200
    S = g.standard_normal((N, 4)).astype(np.float32)
    mean = 0.5 * g.standard_normal((N, 1)); std = 0.3 + 0.4 * g.random((N, 1))
    A = mean + std * g.standard_normal((N, 1)); MU = np.concatenate([mean, std], axis=1)
    R = g.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
    L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * g.standard_normal(N)).astype(np.float32), terminated=int(e % 3 == 0), tag=e)
L.initialize(); L.step(20); L.sync()
t0 = time.perf_counter(); L.step(n); L.sync(); dt = time.perf_counter() - t0
KIND = os.environ.get("KIND", "lstm")
fl = 0
for nin in (4, nC):
    fl += 6.0 * (nin + nC) * (4 if KIND == "lstm" else 2) * nC
print(KIND.upper() + " 2 x %d cells, batch 128, 17 window steps: %.1f us per step; %.2f TFLOP/s of the gate products (forward, dX, dW over 128 x 17 rows)" % (
    nC, dt / n * 1e6, fl * 128 * 17 / (dt / n) / 1e12))

import ctypes as C
g2 = api.lib.hl_debug_stamps; g2.restype = C.c_int; g2.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
acc = []
for it in range(30):
    L.step(2); L.sync()
    out = (C.c_longlong * 32)(); assert g2(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc)
d = np.median(np.diff(a[:, 0:6], axis=1), axis=0) * 10
print("lstm_tm_fwd_kernel, last layer, step 5, workgroup (0, 0), ns: requests + A tile staged %d, barrier %d, products %d, join %d, cell + stores %d | total %d" % (tuple(d) + (int(d.sum()),)))
