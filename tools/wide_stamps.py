"""Device time stamps of workgroup (panel 0, tile 1) of the wide fused kernel (fusedw.hip; library built with
HL_EXTRA_FLAGS=-DHL_PANEL_STAMPS) inside replayed steps, on one Humanoid replica's shape (257 states, 17 actions, 2 x 256, batch 32).
usage: panel_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip

which = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
rg = np.random.default_rng(0)
if which == "atari":
    CONV = [(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]
    L = capi.Learner(api, capi.make_config(dimS=7056, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=6, nAppendedObs=3, conv=CONV, hidden=(512,),
                                           nnFunc="Tanh", batchSize=128, maxTotObsNum=20000, gamma=0.99, explNoise=0.05, randSeed=42))
    L.init_weights()
    for e in range(100):
        N = 60
        S = (255 * rg.random((N, 7056))).astype(np.float32); A = rg.integers(0, 6, size=(N, 1)).astype(np.float64) + 0.1
        MU = rg.random((N, 6)) + 0.2; MU /= MU.sum(1, keepdims=True); R = rg.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
        L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * rg.standard_normal(N)).astype(np.float32), terminated=int(e % 2), tag=e)
else:
    rnn = which == "rnn"
    dS, dA = (6, 1) if rnn else (257, 17)
    kw = dict(dimS=dS, dimA=dA, bounded=[1] if rnn else [0] * dA, hidden=(32, 32) if rnn else (256, 256), batchSize=128 if rnn else 32, maxTotObsNum=131072,
              clipImpWeight=4.0 if rnn else (17 / 2.0) ** 0.5)
    if rnn:
        kw.update(adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnBPTTseq=16, nnFunc="Tanh", gamma=0.99)
    L = capi.Learner(api, capi.make_config(randSeed=7, **kw)); L.init_weights()
    for e in range(300):
        N = 200
        S = rg.standard_normal((N, dS)).astype(np.float32)
        mean = 0.5 * rg.standard_normal((N, dA)); std = 0.3 + 0.4 * rg.random((N, dA))
        A = mean + std * rg.standard_normal((N, dA)); MU = np.concatenate([mean, std], axis=1)
        R = rg.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
        L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * rg.standard_normal(N)).astype(np.float32), terminated=int(e % 3 == 0), tag=e)
L.initialize(); L.step(40)
acc = []
for it in range(40):
    L.step(8)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
if True:
    a = np.array(acc)[:, 0:12]
    d = np.diff(a, axis=1) * 10
    names = ["loads + stage", "h1 tile (MFMA, join)", "h1 store, barrier, read-back", "hoisted head terms", "x2 tile + reduce + stores", "panel barrier", "read-back + stage", "output MFMA, reduce, beta", "head fp64", "delta_y3 / delta_x2 panel", "dX tile + epilogue"]
    for nm, v in zip(names, np.median(d, axis=0)):
        print("%-28s %7.0f ns" % (nm, v))
    print("total %.0f ns" % (np.median(a[:, 11] - a[:, 0]) * 10))
    sys.exit(0)
