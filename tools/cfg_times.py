"""us per gradient step of the other BASELINE.json configurations on one MI355X (synthetic replays, graph replay):
cfg 1 cart_pole_cpp + VRACER.json, cfg 3 one Humanoid replica of 8 (batch 32), cfg 4 RACER_RNN.json (LSTM 2x32, BPTT 16) and the
same with MGU layers, cfg 5 RACER_atari.json.  usage: cfg_times.py [steps]; prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench

if __name__ == "__main__":
    import torch  # noqa: F401
    from smarties_amd import load_hip
    out = bench.other_configs(load_hip(), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    print(json.dumps(out))
