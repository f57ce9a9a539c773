#!/bin/bash
# SQ counters of the RACER_atari step kernels (eager launches): where do the wavefronts spend their cycles?
OUT=/root/repo/gpurun_out/pmc_atari; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SMARTIES_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o r -- python /root/repo/tools/atari_time.py 60 > $OUT/sq.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('/root/repo/gpurun_out/pmc_atari/sq/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('hl::', '')
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_VALU_MFMA_BUSY_CYCLES']
print('%-44s %6s ' % ('kernel', 'n') + ' '.join('%12s' % n[3:15] for n in names))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    if 'conv' in k or 'gemm' in k or 'head' in k or 'stack' in k:
        print('%-44s %6d ' % (k[:44], len(d.get('SQ_WAVE_CYCLES', []))) + ' '.join('%12.0f' % (sum(d.get(n, [0])) / max(1, len(d.get(n, [1])))) for n in names))
import json
ks = []
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    if not ('conv' in k or 'gemm' in k or 'head' in k or 'stack' in k or 'dw_wide' in k): continue
    e = {'kernel': k, 'dispatches': len(d.get('SQ_WAVE_CYCLES', []))}
    for n in names: e[n] = int(sum(d.get(n, [0])) / max(1, len(d.get(n, [1]))))
    wc = max(1, e['SQ_WAVE_CYCLES'])
    e['frac_wait_any'] = round(e['SQ_WAIT_ANY'] / wc, 3); e['frac_issue_stall'] = round(e['SQ_WAIT_INST_ANY'] / wc, 3); e['frac_active'] = round(e['SQ_ACTIVE_INST_ANY'] / wc, 3)
    ks.append(e)
json.dump({'note': 'SQ counters per dispatch (averages over the dispatches) of the RACER_atari.json step, eager launches: SMARTIES_HIP_NO_GRAPH=1 rocprofv3 --pmc ' + ' '.join(names) + ' --kernel-trace -- python tools/atari_time.py 60 (tools/pmc_atari.sh). SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles summed over wavefronts, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md). WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls.', 'kernels': ks},
          open('/root/repo/gpurun_out/pmc_atari/summary.json', 'w'), indent=1)
PY
rm -rf $OUT/sq/*kernel_trace*
