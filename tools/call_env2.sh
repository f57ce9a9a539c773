#!/bin/bash
# the round driver's protocol (bench.py --steps 20 --warmup 5) under launch / kernarg / queue settings of the HIP runtime
cd "$(dirname "$0")/.."
export PYTHONPATH=.
run() { for i in 1 2; do env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('%-50s %.3f us/step  again %.3f  sustained %.3f' % ('$*', d['ms_per_step']*1e3, d['diagnostics']['same_call_again_ms_per_step']*1e3, d['diagnostics']['sustained_ms_per_step_2000_steps']*1e3))
except Exception as e: print('$*', 'failed', e)"; done; }
run X=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run AMD_DIRECT_DISPATCH=0
run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1
