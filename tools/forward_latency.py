"""Latency of hl_forward (RACER::selectAction's network outputs for raw states, Learners/RACER.cpp:30-59): what an env-service
thread pays per agent step, at cfg-NS and for a batch of agents."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bench
from smarties_amd import capi, load_hip
api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG)); L.init_weights()
for e in range(200):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize(); L.step(10); L.sync()
g = np.random.default_rng(0)
for n in (8, 1, 2, 4, 8, 1, 64, 256):
    st = g.standard_normal((n, 17)).astype(np.float32)
    for _ in range(20): L.forward(st)
    t0 = time.perf_counter()
    for _ in range(500): L.forward(st)
    dt = (time.perf_counter() - t0) / 500
    print("hl_forward of %3d states: %.1f us per call" % (n, dt * 1e6))
# while the learner trains (steps enqueued on the same stream)
st = g.standard_normal((1, 17)).astype(np.float32)
L.step(2000)
t0 = time.perf_counter(); L.forward(st); dt = time.perf_counter() - t0
print("hl_forward behind 2000 queued steps: %.1f us" % (dt * 1e6))
# recurrent acting: the agent's last 17 states through two LSTM layers of 32 cells
from oracle_api import fill_synth, synth_cfg
cfg = dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144, randSeed=1, gamma=0.99,
           adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1)
R = capi.Learner(api, capi.make_config(**cfg)); R.init_weights()
fill_synth(R, synth_cfg(seed=3, dimS=4, dimA=1, lenMin=100, lenMax=300, pTerm=0.7), 50); R.initialize()
for T in (1, 17):
    st = g.standard_normal((T, 4)).astype(np.float32)
    for _ in range(300): R.forward_sequence(st)
    t0 = time.perf_counter()
    for _ in range(500): R.forward_sequence(st)
    print("hl_forward_sequence of %2d steps (LSTM 2x32): %.1f us per call" % (T, (time.perf_counter() - t0) / 500 * 1e6))
