cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r05w; rm -rf $O; mkdir -p $O
for k in lstm mgu; do for n in 128 256; do
  (cd /tmp && export TMPDIR=/tmp && KIND=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/wide_${k}_$n -- python /root/repo/tools/lstm_wide_time.py $n 2>&1 | grep "cells") | tee -a $O/wide_time.txt
  cp $(ls $O/wide_${k}_$n/*/*kernel_stats.csv | head -1) $O/wide_${k}_2x${n}_kernel_stats.csv; rm -rf $O/wide_${k}_$n
done; done
for k in lstm mgu; do for n in 128 256; do KIND=$k timeout 200 python tools/lstm_wide_time.py $n 2>&1 | tail -1; done; done | tee $O/wide_time_untraced.txt
timeout 300 python tools/glider_time.py 2>&1 | tail -1 | tee $O/glider_time.txt
