"""What the round driver times (`bench.py --steps 20 --warmup 5`): the first 20-step call of a fresh learner, then the same
call again and again -- separates first-launch effects of the replayed graphs from the steady per-call cost."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from smarties_amd import capi, load_hip

api = load_hip()
L = capi.Learner(api, capi.make_config(**bench.CFG))
L.init_weights()
for e in range(bench.N_EPISODES):
    L.append_episode(**bench.synthetic_episode(np, e))
L.initialize()


def barrier():
    torch.cuda.synchronize(); L.sync()


L.step(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
barrier()
for i in range(6):
    t0 = time.perf_counter(); L.step(20); L.sync(); barrier(); dt = time.perf_counter() - t0
    print("call %d: %.1f us (%.2f per step)" % (i, dt * 1e6, dt * 1e6 / 20))
