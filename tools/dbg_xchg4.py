"""debug: sequences of replica runs of different shapes in ONE process (spec = shape:mode:nr:batch:calls), X (exchange) against H (host sums)"""
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
for spec in sys.argv[1:]:
    shape, mode, nr, batch, calls = spec.split(":")
    nr, batch, calls = int(nr), int(batch), [int(x) for x in calls.split(",")]
    if shape == "ns":
        cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=batch, maxTotObsNum=65536, randSeed=42)
        sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3); eps = 40
    else:
        cfg_kw = dict(dimS=257, dimA=17, hidden=(256, 256), nnFunc="SoftSign", batchSize=batch, maxTotObsNum=65536, randSeed=9)
        sc = synth_cfg(seed=13, dimS=257, dimA=17, lenMin=30, lenMax=120, pTerm=0.3); eps = 30
    os.environ["SMARTIES_HIP_NO_PUSH"] = "1" if mode == "unpushed" else "0"
    X = t6._replicas(api, cfg_kw, sc, nr, eps * nr, True)
    H = t6._replicas(api, cfg_kw, sc, nr, eps * nr, False)
    done = 0
    for n in calls:
        t6._both(X, lambda L: (L.step(n), L.sync()))
        for _ in range(n):
            t6._host_step(H)
        done += n
        bad = False
        for r in range(nr):
            for name, a, b in zip(("W", "M1", "M2"), X[r].get_params(), H[r].get_params()):
                d = np.nonzero(a != b)[0]
                if d.size:
                    bad = True
                    if r == 0 or name == "M1":
                        print(spec[:12], "after", done, "rank", r, name, "differ:", d.size, "max %.3e" % float(np.abs(a - b).max()), "at", d[:8].tolist(), "X", a[d[:3]].tolist(), "H", b[d[:3]].tolist())
        same = all(np.array_equal(X[0].get_params()[0], X[r].get_params()[0]) for r in range(1, nr))
        print(spec[:12], "after", done, "steps:", "MISMATCH" if bad else "equal", "| replicas identical:", same, "| beta", X[0].scalars().beta, H[0].scalars().beta, "far", X[0].scalars().nFarPolicySteps, H[0].scalars().nFarPolicySteps, flush=True)
        if bad:
            break
    for L in X + H:
        L.close()
