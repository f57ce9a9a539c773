"""Device time stamps of the folded weight-gradient launch of a loopback replica (library built with HL_EXTRA_FLAGS=-DHL_FOLD_STAMPS;
see tools/replica_loopback.py for what loopback means)."""
import os, sys, ctypes as C
os.environ["STEPS"] = "8"; os.environ["SMARTIES_HIP_FOLD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
src = open(os.path.join(ROOT, "tools", "replica_loopback.py")).read().split("L.step(64); L.sync()")[0]
exec(src)
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
gk = api.lib.hl_debug_graph_kernels; gk.restype = C.c_int64; gk.argtypes = [C.c_void_p, C.c_int32]
print('kernel nodes of the 8-step graph:', gk(L.h, 8), flush=True)
L.step(64); L.sync()
acc = []
for it in range(30):
    L.step(8); L.sync()
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
print('raw', acc[-1][:16]); a = np.array(acc[3:]).astype(np.float64)
t0 = a[:, 0]
names = {0: "rider entry", 1: "rider bookkeeping done", 2: "rider arrived", 3: "tile 40 entry", 4: "tile 40 computed + pushed", 5: "tile 40 arrived", 14: "last tile of the grid done",
         6: "chunk 0 entry", 7: "chunk 0: own producers all arrived", 8: "chunk 0: peers' stamps seen", 9: "chunk 0: summed", 15: "chunk 0: every chunk's peers have arrived", 10: "chunk 0: Adam applied",
         11: "chunk 0: fence + barrier", 12: "closing workgroup: before the bookkeeping", 13: "closing workgroup: done"}
for i in sorted(names, key=lambda i: np.median(a[:, i] - t0)):
    print("%-48s %8.2f us after the rider's entry" % (names[i], np.median(a[:, i] - t0) / 100.0))
sys.stdout.flush(); os._exit(0)
