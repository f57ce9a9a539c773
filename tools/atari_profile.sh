#!/bin/bash
# rocprofv3 kernel stats of the RACER_atari.json shape (tools/atari_time.py) and of the Humanoid-replica shape on the generic path
# (tools/generic_time.py) into gpurun_out/<tag>
TAG=${1:-atari}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o atari -- python /root/repo/tools/atari_time.py 400 > $OUT/atari_time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o generic -- python /root/repo/tools/generic_time.py 32 > $OUT/generic_time.txt 2>&1
rm -f $OUT/*_kernel_trace.csv
grep "replayed\|us per step" $OUT/atari_time.txt $OUT/generic_time.txt
