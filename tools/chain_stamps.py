"""(chain_stamps.py: the same run with the device time stamps of gemm16.hip: step_chain_kernel; library built with HL_EXTRA_FLAGS=-DHL_CHAIN_STAMPS)
settings/RACER_glider.json shape (RACER, Gaussian advantage, 3 x 128, batch 256): us per replayed step and, under rocprofv3, its kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip
api = load_hip()
kw = dict(dimS=10, dimA=1, bounded=[1], hidden=(128, 128, 128), nnFunc="Tanh", batchSize=256, maxTotObsNum=524288,
          gamma=1.0, adv_kind=capi.ADV_GAUSSIAN, epsAnneal=2e-7, nnLambda=1e-6, penalTol=0.05, clipImpWeight=1.0)
g = np.random.default_rng(5)
L = capi.Learner(api, capi.make_config(randSeed=7, **kw)); L.init_weights()
dS, dA = 10, 1
for e in range(400):
    N = 200
    S = g.standard_normal((N, dS)).astype(np.float32)
    mean = 0.5 * g.standard_normal((N, dA)); std = 0.3 + 0.4 * g.random((N, dA))
    A = mean + std * g.standard_normal((N, dA)); MU = np.concatenate([mean, std], axis=1)
    R = g.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
    L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * g.standard_normal(N)).astype(np.float32), terminated=int(e % 3 == 0), tag=e)
L.initialize(); L.step(64); L.sync()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.perf_counter(); L.step(n); L.sync(); dt = time.perf_counter() - t0
print("glider shape: %.1f us per step" % (dt / n * 1e6))
import ctypes as C
g2 = api.lib.hl_debug_stamps; g2.restype = C.c_int; g2.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
acc = []
for it in range(30):
    L.step(3); L.sync()
    out = (C.c_longlong * 32)(); assert g2(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc)
fw = np.median(np.diff(a[:, 0:7], axis=1), axis=0) * 10
bw = np.median(np.diff(a[:, [6, 16, 17, 18, 19, 20]], axis=1), axis=0) * 10
print("step_chain_kernel, workgroup (panel 0, tile 0), ns: forward (tile, barrier) x 3:", fw.astype(int).tolist(), "| head", int(bw[0]), "| (barrier, dX tile) x 2:", bw[1:].astype(int).tolist(),
      "| total", int(np.median(a[:, 20] - a[:, 0]) * 10))
