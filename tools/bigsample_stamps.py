"""Device time stamps of the large-batch sampler's phases (sample.hip: big_sample_kernel); library built with
HL_EXTRA_FLAGS=-DHL_BIGSAMPLE_STAMPS.  usage: bigsample_stamps.py [batch]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=B, maxTotObsNum=1048576))
L.init_weights(); fill_synth(L, sc, 400 if B <= 2048 else 2500); L.initialize(); L.step(10)
acc = []
for it in range(20):
    L.step(2); L.sync()
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc)
names = ["draws", "sort + unique", "redraw rounds", "Adam draws", "index -> (episode, step)"]
d = np.diff(a[:, :6], axis=1) * 10
e = np.diff(a[:, 8:13], axis=1) * 10
print("   bucket sort (last call): count %d, scan %d, scatter %d, rank %d" % tuple(np.median(e, axis=0)))
print("batch %d, big_sample_kernel, ns (median of 20):" % B, ", ".join("%s %d" % (n, v) for n, v in zip(names, np.median(d, axis=0))), "| total", int(np.median(d.sum(axis=1))))
