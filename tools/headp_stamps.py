"""Device time stamps of the first panel's workgroup of the panel head kernel (headp.hip) on the RACER_RNN.json shape (library built
with HL_EXTRA_FLAGS=-DHL_HEAD_STAMPS): entry, loads issued, panel staged, hoisted head terms, barrier, output layer, head math, barrier,
back-propagation stored."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import fill_synth, synth_cfg
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
KIND = sys.argv[1] if len(sys.argv) > 1 else "lstm"
cfg = dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144, randSeed=1, gamma=0.99,
           adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM if KIND == 'lstm' else capi.NN_MGU, nnLambda=1e-6, explNoise=0.1)
L = capi.Learner(api, capi.make_config(**cfg)); L.init_weights()
fill_synth(L, synth_cfg(seed=3, dimS=4, dimA=1, lenMin=100, lenMax=300, pTerm=0.7), 400)
L.initialize(); L.step(200); L.sync()
acc = []
for it in range(30):
    L.step(8)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
d = np.diff(np.array(acc)[:, 0:9], axis=1) * 10
names = ["loads issued", "panel staged", "hoisted terms", "barrier", "output layer (+ barriers)", "head math", "barrier", "delta_y stored"]
for nm, v in zip(names, np.median(d, axis=0)): print("%-28s %6.0f ns" % (nm, v))
print("total %.0f ns" % np.median(d.sum(axis=1)))
