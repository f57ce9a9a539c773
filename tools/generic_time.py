"""Per-kernel times of the generic (five-launch) path on a Humanoid-like shape: usage generic_time.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from smarties_amd import capi, load_hip
api = load_hip()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dS, dA = 257, 17
kw = dict(dimS=dS, dimA=dA, bounded=[0] * dA, hidden=(256, 256), batchSize=B, maxTotObsNum=131072, clipImpWeight=(17 / 2.0) ** 0.5)
g = np.random.default_rng(5)
L = capi.Learner(api, capi.make_config(randSeed=7, **kw)); L.init_weights()
for e in range(300):
    N = 200
    S = g.standard_normal((N, dS)).astype(np.float32)
    mean = 0.5 * g.standard_normal((N, dA)); std = 0.3 + 0.4 * g.random((N, dA))
    A = mean + std * g.standard_normal((N, dA)); MU = np.concatenate([mean, std], axis=1)
    R = g.standard_normal(N); R[0] = 0; A[-1] = 0; MU[-1] = 0
    L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * g.standard_normal(N)).astype(np.float32), terminated=int(e % 3 == 0), tag=e)
L.initialize(); L.step(64); L.sync()
t0 = time.perf_counter(); L.step(2000); L.sync(); dt = time.perf_counter() - t0
print("batch %d: %.1f us per step" % (B, dt / 2000 * 1e6))
for pid, name in ((21, "fwd0+riderA"), (22, "fwd1+riderB"), (23, "head+riderC"), (24, "dx+post"), (25, "dw+adam"), (12, "empty")):
    try: print("  %-14s %.2f us" % (name, L.kernel_profile(pid, 200)))
    except Exception as e: print("  %-14s n/a %s" % (name, e))
