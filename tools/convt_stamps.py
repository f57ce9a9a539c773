"""Device time stamps of the first row's workgroup in the sample-resident convolution kernels (convt.hip) on the RACER_atari.json
shape; library built with HL_EXTRA_FLAGS=-DHL_CONVT_STAMPS.  usage: convt_stamps.py [n stamps]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
CONV = [(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L = capi.Learner(api, capi.make_config(dimS=7056, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=6, nAppendedObs=3, conv=CONV, hidden=(512,),
                                       nnFunc="Tanh", batchSize=128, maxTotObsNum=20000, gamma=0.99, explNoise=0.05, randSeed=42))
L.init_weights()
rg = np.random.default_rng(0)
for e in range(100):
    N = 60
    S = (255 * rg.random((N, 7056))).astype(np.float32); A = rg.integers(0, 6, size=(N, 1)).astype(np.float64) + 0.1
    MU = rg.random((N, 6)) + 0.2; MU /= MU.sum(1, keepdims=True); R = rg.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
    L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * rg.standard_normal(N)).astype(np.float32), terminated=int(e % 2), tag=e)
L.initialize(); L.step(20)
acc = []
for it in range(30):
    L.step(8)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc)
if len(sys.argv) > 2 and sys.argv[2] == "dw":
    print("conv_dw_dense launch, ns after its first workgroup's start: last end of [rider, row blocks, layers 1-3, dense tiles] =",
          [int(np.median(a[:, 20 + f] - a[:, 19]) * 10) for f in range(4)])
if len(sys.argv) > 2 and sys.argv[2] == "fwd":
    print("conv_fwd_atari_kernel, row 0 (ns): stage + tables %d | layer 1 to its join's barrier %d + %d | layer 2 %d + %d | layer 3 join %d" % tuple(np.median(np.diff(a[:, 24:31], axis=1), axis=0) * 10))
d = np.diff(a[:, 0:n], axis=1) * 10
print("ns between stamps (median of 30):", " ".join("%d" % v for v in np.median(d, axis=0)), "| total", int(np.median(d.sum(axis=1))))
