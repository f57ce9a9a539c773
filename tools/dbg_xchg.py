"""debug: where does the un-pushed exchange differ from the host-formed sum at cfg-NS? (python tools/dbg_xchg.py mode:nr:steps[:close] ...)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "30000")
import numpy as np
import torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
import test_hip_r6 as t6
api = load_hip()
cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
EPS = 40
if os.environ.get("SHAPE") == "humanoid":
    cfg_kw = dict(dimS=257, dimA=17, hidden=(256, 256), nnFunc="SoftSign", batchSize=int(os.environ.get("BATCH", "256")), maxTotObsNum=65536, randSeed=9)
    sc = synth_cfg(seed=13, dimS=257, dimA=17, lenMin=30, lenMax=120, pTerm=0.3)
    EPS = 30
keep = []
for spec in sys.argv[1:]:
    q = spec.split(":")
    mode, nr, calls = q[0], int(q[1]), [int(x) for x in q[2].split(",")]
    close = len(q) < 4 or q[3] != "keep"
    os.environ["SMARTIES_HIP_NO_PUSH"] = "1" if mode == "unpushed" else "0"
    X = t6._replicas(api, cfg_kw, sc, nr, EPS * nr, True)
    H = t6._replicas(api, cfg_kw, sc, nr, EPS * nr, False)
    done = 0
    for n in calls:
        t6._both(X, lambda L: (L.step(n), L.sync()))
        for _ in range(n):
            t6._host_step(H)
        done += n
        bad = False
        if os.environ.get("DUMP") and spec == sys.argv[-1] and done == calls[0]:
            np.savez(os.environ["DUMP"], X=np.stack(X[0].get_params()), H=np.stack(H[0].get_params()))
        if os.environ.get("CMP") and spec == sys.argv[-1] and done == calls[0]:
            ref = np.load(os.environ["CMP"])
            print("   X vs fresh-process X:", int((np.stack(X[0].get_params()) != ref["X"]).sum()), " H vs fresh-process H:", int((np.stack(H[0].get_params()) != ref["H"]).sum()), " fresh X vs fresh H:", int((ref["X"] != ref["H"]).sum()))
        for r in range(nr):
            for name, a, b in zip(("W", "M1", "M2"), X[r].get_params(), H[r].get_params()):
                d = np.nonzero(a != b)[0]
                if d.size:
                    bad = True
                    runs = np.split(d, np.nonzero(np.diff(d) > 1)[0] + 1)
                    print("   nan in X:", int(np.isnan(a).sum()), "nan in H:", int(np.isnan(b).sum()))
                    print(spec, "after", done, "rank", r, name, "differ:", d.size, "max", float(np.abs(a - b).max()), "runs:", [(int(z[0]), int(z[-1])) for z in runs[:10]], len(runs))
        print(spec, "after", done, "steps:", "MISMATCH" if bad else "equal", "beta", X[0].scalars().beta == H[0].scalars().beta, flush=True)
        if bad:
            break
    if close:
        for L in X + H:
            L.close()
    else:
        keep.append((X, H))
