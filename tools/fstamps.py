import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench, numpy as np
from smarties_amd import capi, load_hip
api = load_hip()
g = api.lib.hl_debug_stamps; g.restype = C.c_int; g.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
CFG = dict(bench.CFG)
if len(sys.argv) > 1: CFG['batchSize'] = int(sys.argv[1])     # (with -DHL_FSTAMP_PANEL=p: a workgroup in the crowd)
L = capi.Learner(api, capi.make_config(**CFG)); L.init_weights()
for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()
acc = []
for it in range(40):
    L.step(8)
    out = (C.c_longlong * 32)(); assert g(L.h, out) == 0
    acc.append(np.array(list(out), dtype=np.int64))
a = np.array(acc[5:])
d = np.diff(a[:, 0:14], axis=1) * 10   # ns
names = ['loads+stage', 'h1', 'x2 mma+red', 'epi+st issue', 'precompute', 'waitcnt', 'barrier', 'readback+stage', 'out mma+red', 'head', 'dx2', 'dx mma+red', 'final store']
med = np.median(d, axis=0)
for nm, v in zip(names, med): print('%-16s %7.0f ns' % (nm, v))
print('total', med.sum())
m=np.median(a,axis=0)
print('h1 detail: start->loop end %d ns, epilogue %d ns, sync+reads %d ns' % ((m[14]-m[1])*10, (m[15]-m[14])*10, (m[2]-m[15])*10))
# when the last workgroup of each kind finished, relative to the entry of the stamped workgroup (same launch)
e = a[:, 31]
for nm, i in (("stamped workgroup (panel 0, tile 1)", 13), ("last ordinary panel workgroup", 20), ("last next-state-row panel workgroup", 21), ("sampler rider (draws, sort)", 22), ("far-policy / beta rider", 23)):
    print("%-40s ends %6.0f ns after the entry stamp" % (nm, np.median(a[:, i] - e) * 10))
