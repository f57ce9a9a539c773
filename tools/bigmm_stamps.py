"""Device time stamps of one tile's workgroup in big_mm_kernel (the K = 256 forward product of the bench network at a large batch);
library built with HL_EXTRA_FLAGS=-DHL_BIGMM_STAMPS.  usage: bigmm_stamps.py [batch]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=B, maxTotObsNum=1048576))
L.init_weights(); fill_synth(L, sc, 400 if B <= 2048 else 2500); L.initialize(); L.step(20)
acc = []
for it in range(30):
    L.step(4)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc)
names = ["entry -> row count", "first slice staged", "main loop", "tile to LDS", "epilogue", ]
idx = [0, 1, 2, 3, 4, 6]
d = np.diff(a[:, idx], axis=1) * 10
print("batch %d, workgroup 40 of the K = 256 forward product, ns (median of 30):" % B, ", ".join("%s %d" % (n, v) for n, v in zip(names, np.median(d, axis=0))),
      "| total", int(np.median(d.sum(axis=1))), "| last workgroup of the launch ends %d ns after this one's entry" % int(np.median(a[:, 10] - a[:, 0]) * 10))
