#!/bin/bash
# PMC passes of the step kernels (eager launches: counter collection stalls under graph replay), one counter per run
OUT=/root/repo/gpurun_out/pmc3
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  SMARTIES_HIP_NO_GRAPH=1 timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o r -- python /root/repo/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs --no-diagnostics --no-roofline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob('/root/repo/gpurun_out/pmc3/%s/*counter_collection.csv' % C)
    if not fs: print(C, 'missing'); continue
    acc = collections.defaultdict(list)
    with open(fs[0]) as f:
        for r in csv.DictReader(f):
            acc[r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '').replace('hl::', '')].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res.setdefault(k, {})[C] = (sum(v) / len(v), len(v), max(v))
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', (0, 0, 0))[1])[:10]: print(k, d)
json.dump(res, open('/root/repo/gpurun_out/pmc3/summary.json', 'w'), indent=1)
PY
rm -rf $OUT/FETCH_SIZE/*kernel_trace* $OUT/WRITE_SIZE/*kernel_trace* 2>/dev/null
