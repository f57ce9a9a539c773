#!/bin/bash
# the large-batch step (local batches > 1024: bigmm.hip, headp.hip, post_agg_chunks) per kernel: rocprofv3 --kernel-trace --stats of tools/big_batch.py
cd /root/repo; export PYTHONPATH=.
for B in ${@:-2048 16384}; do
  rm -rf gpurun_out/big$B; mkdir -p gpurun_out/big$B
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/big$B -- python /root/repo/tools/big_batch.py $B 2>&1 | grep batch)
  f=$(ls gpurun_out/big$B/*/*kernel_stats.csv | head -1); python3 tools/kernel_stats.py $f 16
done
