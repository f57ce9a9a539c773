import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
for algo in ("PERerr", "PERrank", "PERseq"):
    L = capi.Learner(api, capi.make_config(dataSamplingAlgo=algo, **bench.CFG)); L.init_weights()
    for e in range(bench.N_EPISODES): L.append_episode(**bench.synthetic_episode(np, e))
    L.initialize(); L.step(5); L.sync()
    t0 = time.perf_counter(); L.step(50); L.sync(); dt = (time.perf_counter() - t0) / 50
    print("%s on 1M transitions: %.2f ms per step" % (algo, dt * 1e3)); L.close()
