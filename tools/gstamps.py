import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench, numpy as np
from smarties_amd import capi, load_hip
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.step(64)
acc = []
for it in range(30):
    L.step(64)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc[3:]).astype(np.float64)
d = np.diff(a[:, 24:29], axis=1) * 10
print('dW tile: loads+stage %d, mfma %d, red+sync %d, adam+store %d ns' % tuple(np.median(d, axis=0)))
# Library built with HL_EXTRA_FLAGS="-DHL_TAIL_STAMPS" (K2 and rider stamps; the K1 phase stamps are tools/fstamps.py with -DHL_FSTAMPS).
# Slots: 0..9 sampler rider, 10/11 far-policy count + beta rider of K1 (start, count done), 12 most rounds of the count's fixed point,
# 13 a main workgroup of K1 has beta, 16..20 bookkeeping rider (written by the steps INSIDE a call: POST_DEFER), 21..23 gather helper 0,
# 24..30 a dW tile, 29 K2 entry, 31 K1 entry.  The last step of every call overwrites the slots that are not restricted to deferred steps.
m = np.median(a, axis=0)
print('dW tile start -> end: %d ns' % ((m[28] - m[24]) * 10))
p = np.diff(a[:, 16:21], axis=1) * 10
print('bookkeeping rider inside a call: start -> end %d ns (scalar loads -> first barrier %d, aggregate loop %d, barrier %d, end phase %d)' %
      (((m[20] - m[16]) * 10,) + tuple(np.median(p, axis=0))))
print('far-policy count: most rounds of the fixed point in any step so far:', sorted(set(int(v) for v in a[:, 12])))
print('K2 entry (block 73) -> tile (40) select done %d ns ; select -> loads issued %d ns ; issued -> staged + sync %d ns' % ((m[24]-m[29])*10, (m[30]-m[24])*10, (m[25]-m[30])*10))
print('K2 riders rel. K2 entry (ns): sampler rider start %d, bookkeeping arrays written %d ; gather helper 0: start %d, search done %d, gathered %d ; tile (40) end %d' %
      tuple((m[i]-m[29])*10 for i in (0, 9, 21, 22, 23, 28)))
print('far-policy count + beta in K1, rel. K1 entry (ns): rider start %d, count done %d ; a main workgroup (panel 0, tile 1) has beta at %d' %
      ((m[10]-m[31])*10, (m[11]-m[31])*10, (m[13]-m[31])*10))
print('K1 entry -> K2 entry %d ns (the other way round: the step time minus this)' % ((m[29]-m[31])*10))
