#!/bin/bash
# the fused two-kernel step at batches between 256 and 1024: us per step and per kernel (rocprofv3 --kernel-trace --stats)
cd /root/repo; export PYTHONPATH=.
for B in ${@:-512 1024}; do
  rm -rf gpurun_out/big; mkdir -p gpurun_out/big
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/big -- python /root/repo/tools/big_batch.py $B 2>&1 | grep batch)
  f=$(ls gpurun_out/big/*/*kernel_stats.csv | head -1); python3 tools/kernel_stats.py $f 8 | grep "fused_fwd\|dw_table"
done
