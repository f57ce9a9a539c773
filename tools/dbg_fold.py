"""debug: after ONE folded step of NR replicas, compare every window slot with the sender's local gradient (host-exchange twins)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "30000")
import numpy as np
import torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, synth_episode
import test_hip_r6 as t6
hip = C.CDLL("libamdhip64.so")
api = load_hip()
nr = int(os.environ.get("NR", "8"))
cfg_kw = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
Ls = []
for r in range(nr):
    L = capi.Learner(api, capi.make_config(n_ranks=nr, rank=r, **cfg_kw)); L.init_weights()
    for e in range(r, 40 * nr, nr):
        L.append_episode(**synth_episode(sc, e))
    Ls.append(L)
w0 = Ls[0].get_params()[0]
for L in Ls:
    w, m1, m2 = L.get_params(); L.set_params(w0, m1, m2); L.initialize()
handles = [L.xchg_export() for L in Ls]
info = [(int.from_bytes(bytes(hd)[64:72], "little"), int.from_bytes(bytes(hd)[80:88], "little")) for hd in handles]
t6._both(Ls, lambda L: L.xchg_connect(handles))
X = Ls
H = t6._replicas(api, cfg_kw, sc, nr, 40 * nr, False)
for L in X + H:
    L.set_tap(True)
for L in H:
    L.step_begin()
gs = [L.grad_fetch() for L in H]
t6._both(X, lambda L: (L.step(1), L.sync()))
n = gs[0].size
from smarties_amd.capi import TAP_FLAT, TAP_STATE, TAP_OUTPUT, TAP_OUTGRAD, TAP_FAR
for r in range(nr):
    print("replica", r, " ".join("%s:%s" % (nm, "same" if np.array_equal(X[r].readback(t), H[r].readback(t)) else "DIFF") for nm, t in (("flat", TAP_FLAT), ("state", TAP_STATE), ("O", TAP_OUTPUT), ("outgrad", TAP_OUTGRAD), ("far", TAP_FAR))),
          "beta", X[r].scalars().beta, H[r].scalars().beta)
R = nr
slotsOffset = (2 * R * 64 * 8 + 255) & ~255
for r in range(min(nr, int(os.environ.get("SHOW", "2")))):
    addr, nbytes = info[r]
    buf = np.zeros(nbytes, np.uint8)
    assert hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(addr), C.c_size_t(nbytes), 2) == 0
    slotBytes = (nbytes - slotsOffset) // (2 * R)
    for sender in range(nr):
        for par in (1,):
            s = buf[slotsOffset + (par * R + sender) * slotBytes:][:n * 4].view(np.float32)
            d = np.nonzero(s != gs[sender][:n])[0]
            cols = sorted(set(((d[(d >= 4608) & (d < 70144)] - 4608) % 256) // 16)) if d.size else []
            print("window of", r, "slot of sender", sender, "differs from that rank's local gradient at", d.size, "first", d[:4], "column tiles of W1 affected:", cols, "zeros there:", int((s[d] == 0).sum()) if d.size else 0)
g = gs[0].copy()
for q in gs[1:]:
    g = (g + q).astype(np.float32)
for r in range(min(nr, 2)):
    xg = X[r].readback(capi.TAP_GRADSUM); m = min(xg.size, g.size)
    d = np.nonzero(xg[:m] != g[:m])[0]
    print("X", r, "summed G vs host sum: differ", d.size)
