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
# relation to the fused kernel of the same step (stamp 13 = end of K1 tile (panel 0, n 1)), post stamps 16..20
m = np.median(a, axis=0)
print('K1 end -> dW tile start: %d ns' % ((m[24] - m[13]) * 10))
print('K1 start -> K1 end: %d ns' % ((m[13] - m[0]) * 10))
print('dW tile start -> end: %d ns' % ((m[28] - m[24]) * 10))
print('post: start->end %d ns ; post start rel. dW tile start %d ns' % ((m[20] - m[16]) * 10, (m[16] - m[24]) * 10))
p = np.diff(a[:, 16:21], axis=1) * 10
print('post stamps (ns): scalar loads->init-sync %d, agg-loop %d, sync %d, thread0-rest %d' % tuple(np.median(p, axis=0)))
print('far count: sync -> terms reloaded %d ns ; fixed point %d ns ; -> end of the pass %d ns' % ((m[14]-m[19])*10, (m[15]-m[14])*10, (m[20]-m[15])*10))
print('far count: most rounds in any step so far:', sorted(set(int(v) for v in a[:, 12])))
print('K2 entry(block 73) -> tile(40) select done %d ns ; select -> loads issued %d ns ; issued -> staged+sync %d ns' % ((m[24]-m[29])*10, (m[30]-m[24])*10, (m[25]-m[30])*10))
print('K1 end -> K2 entry %d ns' % ((m[29]-m[13])*10))
print('K2 tile end -> K1 entry %d ns ; K1 entry -> stamp0 %d ns ; K1 stamp0 -> end %d ns' % ((m[31]-m[28])*10 , (m[0]-m[31])*10, (m[13]-m[0])*10))
d = np.diff(a[:, 0:14], axis=1) * 10
names = ['loads+stage', 'h1', 'x2 mma+red', 'epi+st issue', 'precompute', 'waitcnt', 'barrier', 'readback+stage', 'out mma+red', 'head', 'dx2', 'dx mma+red', 'final store']
print(' | '.join('%s %d' % (nm, v) for nm, v in zip(names, np.median(d, axis=0))))
print('K1: tile(0,1) end -> last main WG end %d ns ; -> last rider end %d ns ; last main WG end -> K2 entry %d ns' % ((m[15]-m[13])*10, (m[14]-m[13])*10, (m[29]-m[15])*10))
print('all stamps relative to K2 entry (ns):', ' '.join('%d:%d' % (i, (m[i] - m[29]) * 10) for i in range(32)))
print('K2 riders rel. K2 entry (ns): sampler rider start %d, minibatch published %d ; gather helper 0: start %d, hand-off seen %d, gathered %d ; bookkeeping start %d, end %d ; tile(40) end %d' % tuple((m[i]-m[29])*10 for i in (0, 9, 21, 22, 23, 16, 20, 28)))
print('deferred count + beta in K1, rel. K1 entry (ns): rider start %d, count done %d ; main workgroup (panel 0, tile 1) has beta at %d, present at its first look: %d' % ((m[10]-m[31])*10, (m[11]-m[31])*10, (m[13]-m[31])*10, m[3]))
print('K2 entry -> next K1 entry %d ns ; K1 entry -> K2 entry %d ns' % ((m[31]-m[29])*10, (m[29]-m[31])*10))
