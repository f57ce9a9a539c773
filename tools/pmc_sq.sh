#!/bin/bash
# SQ counters per kernel of any timing tool (eager launches): usage pmc_sq.sh <tag> <python script and arguments>
# e.g. tools/pmc_sq.sh k1_b512 tools/big_batch.py 512     -> gpurun_out/pmc_sq/<tag>.json
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmc_sq; mkdir -p $OUT; rm -rf $OUT/$TAG; cd /tmp; export TMPDIR=/tmp
SMARTIES_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$TAG -o r -- python /root/repo/$@ > $OUT/$TAG.log 2>&1
echo rc=$?
python3 - $OUT/$TAG <<'PY'
import csv, glob, collections, json, sys
d0 = sys.argv[1]
fs = glob.glob(d0 + '/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('hl::', '')
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES']
ks = []
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])))[:8]:
    e = {'kernel': k, 'dispatches': len(d.get('SQ_WAVE_CYCLES', []))}
    for n in names: e[n] = int(sum(d.get(n, [0])) / max(1, len(d.get(n, [1]))))
    wc = max(1, e['SQ_WAVE_CYCLES'])
    e['frac_wait_any'] = round(e['SQ_WAIT_ANY'] / wc, 3); e['frac_issue_stall'] = round(e['SQ_WAIT_INST_ANY'] / wc, 3); e['frac_active'] = round(e['SQ_ACTIVE_INST_ANY'] / wc, 3); e['frac_valu'] = round(e['SQ_ACTIVE_INST_VALU'] / wc, 3)
    ks.append(e)
    print('%-40s n %5d ' % (k[:40], e['dispatches']) + ' '.join('%s %d' % (n[3:], e[n]) for n in names), '| wait %.2f stall %.2f active %.2f valu %.2f' % (e['frac_wait_any'], e['frac_issue_stall'], e['frac_active'], e['frac_valu']))
json.dump({'note': 'SQ counters per dispatch (averages), eager launches; quad-cycles summed over wavefronts for WAVE / WAIT / ACTIVE, cycles for MFMA_BUSY (tools/pmc_sq.sh)', 'kernels': ks}, open(d0 + '.json', 'w'), indent=1)
PY
rm -rf $OUT/$TAG
