"""Stress: the same small discrete-RACER learner built and stepped many times in one process; any run whose end
weights differ bitwise from the first exposes a race in the device path."""
import sys, os, hashlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import fill_synth, synth_cfg
api = load_hip()
kind = sys.argv[1] if len(sys.argv) > 1 else "discrete"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
if kind == "discrete":
    kw = dict(dimS=7, dimA=1, bounded=[0], n_options=5, hidden=(32, 32), nnFunc="SoftSign", batchSize=32, maxTotObsNum=5000,
              randSeed=99, adv_kind=capi.ADV_DISCRETE, clipImpWeight=4.0, outWeightsPrefac=0.1)
elif kind == "gauss":
    kw = dict(dimS=7, dimA=2, bounded=[1, 0], hidden=(32, 32), nnFunc="SoftSign", batchSize=32, maxTotObsNum=5000,
              randSeed=99, adv_kind=capi.ADV_GAUSSIAN, clipImpWeight=4.0)
else:
    kw = dict(dimS=7, dimA=2, bounded=[1, 0], hidden=(32, 32), nnFunc="SoftSign", batchSize=32, maxTotObsNum=5000,
              randSeed=99, clipImpWeight=4.0)
sc = synth_cfg(seed=5, dimS=7, dimA=kw["dimA"], lenMin=20, lenMax=41, pTerm=0.5)
seen = {}
for rep in range(reps):
    L = capi.Learner(api, capi.make_config(**kw)); L.init_weights()
    fill_synth(L, sc, 40)
    L.initialize()
    hs = []
    for k in range(20):
        L.step(1)
        L.readback(capi.TAP_FLAT) if False else None
        if os.environ.get("DET_PER_STEP"):
            hs.append(hashlib.md5(L.get_params()[0].tobytes()).hexdigest()[:8])
    w = L.get_params()[0]
    h = hashlib.md5(w.tobytes()).hexdigest()[:10]
    seen.setdefault(h, []).append(rep)
    if hs: seen.setdefault("steps:" + ",".join(hs[:20]), []).append(rep)
    L.close()
print(kind, {k[:40]: (len(v), v[:5]) for k, v in seen.items()})
