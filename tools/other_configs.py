"""us per step of bench.py's other_configs rows whose name contains one of the arguments (all without arguments)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: F401
import bench
from smarties_amd import load_hip
print(json.dumps(bench.other_configs(load_hip(), steps=int(os.environ.get("STEPS", "600")), only=sys.argv[1:] or None), indent=1))
