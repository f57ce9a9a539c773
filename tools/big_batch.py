"""The bench network (17-256-256-7, V-RACER) at batches 256 ... 16384: us per replayed step, TFLOP/s of the 421 376 FLOP per
transition, fraction of the fp32 MFMA peak -- where the step stops being a latency chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch  # noqa: F401
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
api = load_hip()
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
for B in [int(x) for x in (sys.argv[1:] or [256, 1024, 4096, 16384])]:
    nEp = int(os.environ.get('NEP', 0)) or 400 if B <= 2048 else 2500            # (a replay only five times the batch costs the sampler seven redraw rounds per minibatch)
    L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=B, maxTotObsNum=1048576))
    L.init_weights(); fill_synth(L, sc, nEp); L.initialize()
    L.step(40); L.sync()
    n = max(20, 200000 // B)
    t0 = time.perf_counter(); L.step(n); L.sync(); dt = (time.perf_counter() - t0) / n
    tf = 421376.0 * B / dt / 1e12
    print("batch %6d: %8.1f us per step, %6.2f TFLOP/s = %.3f of the fp32 MFMA peak, %.2f M transitions/s" % (B, dt * 1e6, tf, tf / 157.3, B / dt / 1e6))
    L.close()
