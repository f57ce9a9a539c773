#!/bin/bash
# round 6: the bench line (default run + the driver's protocol), rocprofv3 kernel statistics of the bench command, the PMC traffic passes,
# the replica loopback / stamps; results under gpurun_out/r06f (copied to profiles/r06_* by hand)
cd /root/repo; export PYTHONPATH=.
O=gpurun_out/r06f; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep "^{" $O/bench_default.log > $O/bench.json; tail -c 400 $O/bench.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_driver_protocol.json 2>/dev/null; tail -c 300 $O/bench_driver_protocol.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kt -o r -- python /root/repo/bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-other-configs --no-diagnostics > /root/repo/$O/bench_rocprof.log 2>&1)
grep "^{" $O/bench_rocprof.log > $O/bench_under_rocprof.json; cp $(ls $O/kt/*kernel_stats.csv $O/kt/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv; rm -rf $O/kt
head -8 $O/kernel_stats.csv | cut -c1-160
timeout 500 bash tools/pmc3.sh 2>&1 | tail -6; cp gpurun_out/pmc3/summary.json $O/pmc.json
for nr in 2 8; do NR=$nr timeout -k 5 100 python tools/replica_loopback.py 2>&1 | grep loopback; NR=$nr SMARTIES_HIP_NO_PUSH=1 timeout -k 5 100 python tools/replica_loopback.py 2>&1 | grep loopback; NR=$nr SMARTIES_HIP_FOLD=1 timeout -k 5 100 python tools/replica_loopback.py 2>&1 | grep loopback; done | tee $O/replica_loopback.txt
