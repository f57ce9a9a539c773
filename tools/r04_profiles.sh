#!/bin/bash
# everything profiles/r04_* comes from, in one GPU-box session (gpurun -- tools/r04_profiles.sh); results under gpurun_out/r04
cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r04; mkdir -p $O
tools/run_gpu.sh r04 4000 6 > $O/run_gpu.txt 2>&1; tail -25 $O/run_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver_protocol.json 2>/dev/null
tools/atari_profile.sh r04 > $O/atari_profile.txt 2>&1; tail -3 $O/atari_profile.txt
tools/rec_profile.sh r04 > $O/rec_profile.txt 2>&1; grep "per step" $O/rec_profile.txt
tools/pmc3.sh > $O/pmc3.txt 2>&1; tail -4 $O/pmc3.txt; cp gpurun_out/pmc3/summary.json $O/pmc3_summary.json
tools/pmc_atari.sh > $O/pmc_atari.txt 2>&1; tail -16 $O/pmc_atari.txt
HL_EXTRA_FLAGS="-DHL_FSTAMPS" python -c "import __graft_entry__ as g; g.build_hip()" > /dev/null 2>&1; timeout 200 python tools/fstamps.py > $O/fstamps.txt 2>&1; tail -21 $O/fstamps.txt
HL_EXTRA_FLAGS="-DHL_PANEL_STAMPS" python -c "import __graft_entry__ as g; g.build_hip()" > /dev/null 2>&1; timeout 200 python tools/wide_stamps.py > $O/wide_stamps.txt 2>&1; tail -13 $O/wide_stamps.txt
python -c "import __graft_entry__ as g; g.build_hip()" > /dev/null 2>&1
for N in 2 8; do
  SMARTIES_BENCH_PG=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 20 --warmup 5 > $O/bench_gpus${N}_dryrun_one_device.json 2> $O/bench_gpus${N}.err; tail -c 900 $O/bench_gpus${N}_dryrun_one_device.json; echo
done
