#!/bin/bash
# large local batches: us per step and per kernel at B = 2048 / 4096 / 16384 with the generic 16-row-tile launches (SMARTIES_HIP_GENERIC=128)
# and with the kernels of bigmm.hip (3), each under rocprofv3 --kernel-trace --stats
cd /root/repo; export PYTHONPATH=.
for B in 2048 4096 16384; do for m in 0 3; do
  rm -rf gpurun_out/big; mkdir -p gpurun_out/big
  (cd /tmp && export TMPDIR=/tmp && SMARTIES_HIP_GENERIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/big -- python /root/repo/tools/big_batch.py $B 2>&1 | grep batch)
  echo "--- B=$B BIGMM=$m"; f=$(ls gpurun_out/big/*/*kernel_stats.csv | head -1); python3 tools/kernel_stats.py $f 11 | grep -v "ingest\|fillBuffer\|sweep\|step_tail\|stack_gather"
done; done
