#!/bin/bash
# usage: run_gpu.sh <tag> [steps]  -- tests + smoke + bench + rocprof kernel stats into gpurun_out/<tag>
TAG=$1; STEPS=${2:-4000}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | grep -v "^  \|^$\|amdgpu.ids" | tail -${3:-8}
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; grep "^{" $OUT/bench_default.log > $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python /root/repo/bench.py --steps $STEPS --warmup 200 --no-cpu-baseline --no-other-configs --no-diagnostics > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log > $OUT/bench_rocprof.json
rm -f $OUT/r_kernel_trace.csv
python3 - <<PY
import json, csv
for f in ('bench_default.json', 'bench_rocprof.json'):
    try:
        d = json.load(open('$OUT/' + f))
        print(f, 'value %.0f ms/step %.5f' % (d['value'], d['ms_per_step']), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
        r = d['roofline']; print('  dominant', r['kernel'], r['bound'], 'achieved %.3f %s frac %.5f' % (r['achieved'], r['unit'], r['frac']), 'empty', r['empty_launch_us'])
        for k, v in r['step_kernels'].items(): print('   %-26s launch %.2f us frac %.4f' % (k, v['launch_us'], v['frac']))
    except Exception as e: print(f, 'FAILED', e)
rows = list(csv.DictReader(open('$OUT/r_kernel_stats.csv')))
for r in rows[:14]: print('%-60s calls %6s avg %9.1f ns  min %s max %s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
