import sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from smarties_amd import capi, load_hip
api = load_hip()
for use_comm in (0, 1):
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
    if use_comm:
        raw = (C.c_uint8 * 128)(); assert api.fn("comm_unique_id")(raw) == 0
        L.comm_init(bytes(raw))
    L.initialize(); L.step(300); L.sync()
    t0 = time.perf_counter(); L.step(3000); L.sync(); dt = time.perf_counter() - t0
    print('comm attached' if use_comm else 'single replica', '%.2f us/step' % (dt / 3000 * 1e6))
    L.close()
