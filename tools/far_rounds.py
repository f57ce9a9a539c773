"""Rounds of the far-policy count's fixed point over a long run (library built with HL_EXTRA_FLAGS=-DHL_TAIL_STAMPS: the most
rounds any step needed is kept in a debug slot), and the longest 50-step call."""
import sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench, numpy as np
from smarties_amd import capi, load_hip
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.prepare_steps(50)
worst = 0.0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    t0 = time.perf_counter(); L.step(50); L.sync(); worst = max(worst, time.perf_counter() - t0)
out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
print("most rounds of the fixed point in any step: %d ; longest 50-step call %.0f us" % (out[12], worst * 1e6))
