#!/bin/bash
# end of round 5: GPU tests, smoke, the bench line (default run + the driver's protocol), rocprofv3 kernel statistics of the bench command,
# of the large-batch step (tools/big_profile.sh) and of the wide recurrent nets; results under gpurun_out/r05f (copied to profiles/r05_*)
cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r05f; mkdir -p $O
tools/run_gpu.sh r05f 4000 6 > $O/run_gpu.txt 2>&1; tail -30 $O/run_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver_protocol.json 2>/dev/null; tail -c 300 $O/bench_driver_protocol.json; echo
bash tools/big_profile.sh 2048 4096 16384 2>&1 | grep -v "simple_timer\|ingest\|rocclr\|moments\|episode_sweep" > $O/big_profile.txt; cat $O/big_profile.txt
for B in 2048 16384; do cp $(ls gpurun_out/big$B/*/*kernel_stats.csv | head -1) $O/big${B}_kernel_stats.csv; done
for k in lstm mgu; do for n in 128 256; do
  (cd /tmp && export TMPDIR=/tmp && KIND=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/wide_${k}_$n -- python /root/repo/tools/lstm_wide_time.py $n 2>&1 | grep "cells") | tee -a $O/wide_time.txt
  cp $(ls $O/wide_${k}_$n/*/*kernel_stats.csv | head -1) $O/wide_${k}_2x${n}_kernel_stats.csv; rm -rf $O/wide_${k}_$n
done; done
for k in lstm mgu; do for n in 128 256; do KIND=$k timeout 200 python tools/lstm_wide_time.py $n 2>&1 | tail -1; done; done | tee $O/wide_time_untraced.txt
timeout 300 python tools/glider_time.py 2>&1 | tail -1 | tee $O/glider_time.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/glider -- python /root/repo/tools/glider_time.py 2>&1 | grep shape)
cp $(ls $O/glider/*/*kernel_stats.csv | head -1) $O/glider_kernel_stats.csv; rm -rf $O/glider
