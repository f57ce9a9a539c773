#!/bin/bash
# re-takes the parts of profiles/r04_* that the recurrent one-launch step and the B = 1024 bench row changed (the rest: tools/r04_profiles.sh)
cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r04b; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver_protocol.json 2>/dev/null; tail -c 300 $O/bench_driver_protocol.json; echo
tools/rec_profile.sh r04b > $O/rec_profile.txt 2>&1; grep "per step" $O/rec_profile.txt
HL_EXTRA_FLAGS="-DREC_STAMPS" python -c "import __graft_entry__ as g; g.build_hip()" > /dev/null 2>&1; timeout 200 python tools/lstm_wave_stamps.py > $O/lstm_wave_stamps.txt 2>&1; tail -4 $O/lstm_wave_stamps.txt
