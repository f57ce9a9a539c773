"""RACER_atari.json shape (BASELINE config 5) on the device: 84x84 frames x (1 + 3 appended), four SoftSign convolutions,
dense 512 + parametric residual, 6 options, batch 128.  Synthetic replay (episodes of 60 states), us per step of replayed
graphs and HIP-event time per kernel of eager steps.  usage: atari_time.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from smarties_amd import capi, load_hip

CONV = [(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
api = load_hip()
cfg = capi.make_config(dimS=7056, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=6, nAppendedObs=3, conv=CONV, hidden=(512,),
                       nnFunc="Tanh", batchSize=128, maxTotObsNum=20000, gamma=0.99, explNoise=0.05, randSeed=42)
L = capi.Learner(api, cfg)
L.init_weights()
g = np.random.default_rng(0)
N = 60
for e in range(200):
    S = (255 * g.random((N, 7056))).astype(np.float32)
    A = g.integers(0, 6, size=(N, 1)).astype(np.float64) + 0.1
    MU = g.random((N, 6)) + 0.2
    MU /= MU.sum(1, keepdims=True)
    R = g.standard_normal(N); R[0] = 0
    A[-1] = 0; MU[-1] = 0
    V = (0.5 * g.standard_normal(N)).astype(np.float32)
    L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=V, terminated=int(e % 2), tag=e)
L.initialize()
L.step(33); L.sync()
t0 = time.perf_counter(); L.step(steps); L.sync(); dt = time.perf_counter() - t0
print("replayed graphs: %.1f us per step, %.0f transitions/s" % (dt / steps * 1e6, 128 * steps / dt))
api.fn("timing_enable")(L.h, 1)
L.step(50); L.sync()
import ctypes as C
tot = 0.0
for nm in ("step_tail_kernel", "stack_gather", "conv_prep", "conv_fwd0", "conv_fwd1", "conv_fwd2", "conv_fwd3", "gemm16_fwd1", "head_kernel", "gemm16_dx1",
           "conv_dx3", "conv_dx2", "conv_dx1", "conv_back", "conv_fwd_tail", "conv_dw_dense", "conv_dw_all", "conv_dw_rows", "conv_dw", "conv_reduce_adam", "gemm16_dw", "dw_direct", "post_kernel"):
    ms, n = C.c_double(), C.c_int64()
    api.fn("timing_get")(L.h, nm.encode(), C.byref(ms), C.byref(n))
    print("  %-18s %8.2f us  x%d" % (nm, ms.value * 1e3, n.value)); tot += ms.value * 1e3
print("  sum %.1f us" % tot)
api.fn("timing_enable")(L.h, 0)
