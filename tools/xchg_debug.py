import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
import test_hip_parity as T
class A: pass
api = load_hip()
cfg_kw = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)
sc = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)
X = T._xchg_replicas(api, cfg_kw, sc, True)
H = T._xchg_replicas(api, cfg_kw, sc, False)
T._both(X, lambda L: (L.step(1), L.sync()))
for L in H: L.step_begin()
gs = [L.grad_fetch() for L in H]
g = np.sum(gs, axis=0, dtype=np.float32)
gx = [L.grad_fetch() for L in X]
print("summed gradient equal:", [np.array_equal(q, g) for q in gx], "max diff", [float(np.abs(q - g).max()) for q in gx], "norm", float(np.abs(g).max()))
bad = np.nonzero(gx[0] != g)[0]; print("n differing", bad.size, bad[:10], gx[0][bad[:5]], g[bad[:5]], gs[0][bad[:5]], gs[1][bad[:5]])
