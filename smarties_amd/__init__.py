"""smarties_amd -- MI355X-native learner-update hot path of cselab/smarties (V-RACER / ReF-ER).

Only what the path needs: ``csrc/`` (hand-written HIP kernels for gfx950 + the hl_* C-ABI,
built into ``libsmarties_hip.so``) and a ctypes binding (``capi``).  See DESIGN.md.
"""
from .capi import (CApi, Learner, HlConfig, HlError, make_config, load_hip, HIP_LIB)  # noqa: F401
