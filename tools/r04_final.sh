#!/bin/bash
# re-takes what the convolutional step's merged weight-gradient launch and the prioritised samplers' scan changed at the end of round 4
# (the rest of profiles/r04_*: tools/r04_profiles.sh, tools/r04_refresh.sh); results under gpurun_out/r04c
cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r04c; mkdir -p $O
tools/run_gpu.sh r04c 4000 6 > $O/run_gpu.txt 2>&1; tail -25 $O/run_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver_protocol.json 2>/dev/null; tail -c 400 $O/bench_driver_protocol.json; echo
tools/atari_profile.sh r04c > $O/atari_profile.txt 2>&1; tail -3 $O/atari_profile.txt
timeout 300 python tools/per_time.py > $O/per_time.txt 2>&1; grep "ms per step" $O/per_time.txt
