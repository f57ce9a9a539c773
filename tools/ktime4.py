import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from smarties_amd import capi, load_hip
api = load_hip()
f = api.lib.hl_debug_kernel_time; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.step(50)
names = {12:'empty', 0:'sample', 8:'phase A', 9:'phase B', 10:'phase C', 11:'post(AGG|BETA)',
         28:'K1 fused (no rider)', 26:'K1 fused + A,B', 29:'K2 dw+adam (no rider)', 27:'K2 + C + post', 7:'fused step'}
def t(which, variant=0, reps=200):
    us = C.c_double(); rc = f(L.h, which, reps, variant, C.byref(us));
    if rc: return float('nan')
    return us.value
for w in (12, 28, 26, 29, 27, 7): print('%-24s %.2f us' % (names[w], t(w, 0, 64 if w == 7 else 200)))
for v in range(1, 9): print('K1 stop after phase %d: %.2f us' % (v, t(28, v)))
print('K1 full %.2f' % t(28, 0))
