#!/bin/bash
# the driver protocol's call (hl_step(20) + hl_sync + device synchronize) under host-side wait / launch settings of the HIP runtime
# usage (GPU box, library built with HL_EXTRA_FLAGS=-DHL_STEP_STAMPS): tools/call_env.sh [steps per call]
N=${1:-20}
cd "$(dirname "$0")/.."
export PYTHONPATH=.
run() { echo "== $*"; env "$@" python tools/step_stamps.py $N 14 2>&1 | grep -v amdgpu.ids | tail -4; }
run X=1
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run HSA_ENABLE_INTERRUPT=0
run HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
# (ROC_SYSTEM_SCOPE_SIGNAL=0 wedges the first synchronisation on this runtime: not run)
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
