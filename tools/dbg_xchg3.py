"""debug: in the failing sequence, which gradient did the host-exchange replica's Adam launch see -- the stored sum or its own local one?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "30000")
os.environ["SMARTIES_HIP_NO_PUSH"] = "1"
import numpy as np
import torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg
import test_hip_r6 as t6
api = load_hip()
cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
withX = os.environ.get("WITHX", "1") == "1"
for it in range(int(os.environ.get("ITERS", "8"))):
    xmode = os.environ.get("XMODE", "full")
    if not withX:
        X = []
    elif xmode in ("full", "nostep"):
        X = t6._replicas(api, cfg_kw, sc, 2, 80, True)
    else:
        X = t6._replicas(api, cfg_kw, sc, 2, 80, False)
        if xmode == "export":
            hd = [L.xchg_export() for L in X]
    H = t6._replicas(api, cfg_kw, sc, 2, 80, False)
    if withX and xmode == "full":
        t6._both(X, lambda L: (L.step(1), L.sync()))
    for L in H:
        L.step_begin()
    gs = [L.grad_fetch() for L in H]
    g = (gs[0] + gs[1]).astype(np.float32)
    if it == 0:
        gs0 = [q.copy() for q in gs]
    else:
        for r in range(2):
            d = np.nonzero(gs[r] != gs0[r])[0]
            runs = np.split(d, np.nonzero(np.diff(d) > 1)[0] + 1) if d.size else []
            print("iter", it, "rank", r, "LOCAL gradient differs from iteration 0's at", d.size, [(int(z[0]), int(z[-1])) for z in runs[:6]], len(runs))
            for tap, nm in ((capi.TAP_FLAT, "flat"), (capi.TAP_OUTPUT, "O"), (capi.TAP_OUTGRAD, "outgrad")):
                pass
    if withX and xmode == "full":
        for r in range(2):
            xg = X[r].readback(capi.TAP_GRADSUM); n = min(xg.size, g.size)
            print("iter", it, "X", r, "summed G vs H's host sum differ:", int((xg[:n] != g[:n]).sum()))
    cs = np.sum([L.counters_fetch() for L in H], axis=0)
    for L in H:
        L.grad_store(g); L.counters_store(cs); L.step_end()
    for r in range(2):
        m1 = H[r].get_params()[1]
        n = min(m1.size, g.size)
        exp_sum = (0.1 * (g[:n] / 256.0)).astype(np.float32)
        exp_loc = (0.1 * (gs[r][:n] / 256.0)).astype(np.float32)
        bad = np.nonzero(~np.isclose(m1[:n], exp_sum, rtol=1e-5, atol=1e-12))[0]
        isloc = np.isclose(m1[bad], exp_loc[bad], rtol=1e-5, atol=1e-12).sum() if bad.size else 0
        back = H[r].grad_fetch()
        print("iter", it, "rank", r, "M1 not from the stored sum:", bad.size, "of which equal to the LOCAL gradient:", int(isloc), "| G read back != stored:", int((back[:n] != g[:n]).sum()), flush=True)
    for L in X + H:
        L.close()
