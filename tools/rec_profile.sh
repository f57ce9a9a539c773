#!/bin/bash
# rocprofv3 kernel statistics of the recurrent path (RACER_RNN.json shape, LSTM and MGU) into gpurun_out/<tag>
TAG=${1:-rec}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for K in lstm mgu; do
  python /root/repo/tools/lstm_time.py $K 2>&1 | grep "per step" > $OUT/${K}_time.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$K -o r -- python /root/repo/tools/lstm_time.py $K > /dev/null 2>&1
  cp $OUT/$K/r_kernel_stats.csv $OUT/${K}_kernel_stats.csv 2>/dev/null || find $OUT/$K -name "*kernel_stats.csv" -exec cp {} $OUT/${K}_kernel_stats.csv \;
  rm -rf $OUT/$K
  cat $OUT/${K}_time.txt; head -7 $OUT/${K}_kernel_stats.csv
done
