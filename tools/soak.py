import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from smarties_amd import capi, load_hip
api = load_hip()
res = []
for run in range(2):
    L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
    for e in range(5000): L.append_episode(**bench.synthetic_episode(np, e))
    L.initialize()
    t0 = time.perf_counter()
    for chunk in range(10):
        L.step(20000); L.sync()
    dt = time.perf_counter() - t0
    w, m1, m2 = L.get_params(); sc = L.scalars()      # get_scalars also checks the device error flag
    res.append((w.copy(), m1.copy(), sc.beta, sc.nFarPolicySteps, L.get_rng_state().copy()))
    print('run %d: 200000 steps in %.2f s (%.2f us/step), beta %.12g nFar %d |w| %.6f finite %s' % (run, dt, dt / 2e5 * 1e6, sc.beta, sc.nFarPolicySteps, float(np.linalg.norm(w)), bool(np.isfinite(w).all())))
    L.close()
same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2] and res[0][3] == res[1][3] and np.array_equal(res[0][4], res[1][4])
print('two runs of 200k steps bit-identical:', same)
