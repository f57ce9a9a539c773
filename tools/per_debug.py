import sys, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from smarties_amd import capi
from parity import *
api = capi.load_hip()
fx = load_fixture(sys.argv[1] if len(sys.argv) > 1 else "sample_PERrank.bin")
L = capi.Learner(api, fixture_config(fx)); setup_from_fixture(L, fx); L.set_tap(True)
f = api.lib.hl_debug_per_table; f.restype = C.c_int64; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
L.step(1)
prob = np.zeros(4096, np.float32); cp = np.zeros(4096, np.float64)
n = f(L.h, prob.ctypes.data, cp.ctypes.data, 4096)
print("n", n, "prob", prob[:8], prob[n-4:n+1], "cp", cp[:6], cp[n-3:n+1])
s = 0.0
for v in prob[:n]: s += float(v)
q = prob[:n].astype(np.float64) / s
ref = np.zeros(n); a = 0.0
for i in range(n): a = a + q[i]; ref[i] = a
ref[-1] = 1
print("cp equal:", np.array_equal(ref, cp[:n]), np.abs(ref - cp[:n]).max())
print(L.readback(capi.TAP_FLAT)); print(fx["s1_flat"])
print("tag", L.readback(capi.TAP_TAG)); print("fx ", fx["s1_tag"]); print("t  ", L.readback(capi.TAP_TSTEP)); print("fx ", fx["s1_t"])
print([L.episode_info(p) for p in range(6)])
for k in range(2, 6):
    r = L.get_rng_state(); print(k, "rng eq", np.array_equal(r, fx["s%d_rng" % k]))
    L.step(1); print(L.readback(capi.TAP_FLAT)); print(fx["s%d_flat" % k])
