#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of tools/pmc_calib.hip (each moves 256 MiB): the factor between counter and bytes per access pattern
OUT=/root/repo/gpurun_out/pmc_calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 /root/repo/tools/pmc_calib.hip -o /tmp/pmc_calib || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o r -- /tmp/pmc_calib > $OUT/$C.log 2>&1; echo "$C rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob('/root/repo/gpurun_out/pmc_calib/%s/*counter_collection.csv' % C)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    for k, v in acc.items(): res[k][C + '_KB'] = sum(v) / len(v)
MB = 256.0 * 1024
out = {"note": "tools/pmc_calib.sh on MI355X: each kernel moves 262144 KB; ratio = counter (KB) / bytes moved (KB)", "kernels": {}}
for k, d in sorted(res.items()):
    out["kernels"][k] = {kk: round(v, 1) for kk, v in d.items()}
    out["kernels"][k].update({"fetch_ratio": round(d.get("FETCH_SIZE_KB", 0) / MB, 3), "write_ratio": round(d.get("WRITE_SIZE_KB", 0) / MB, 3)})
    print("%-28s FETCH %.0f KB (x%.3f of 256 MiB)   WRITE %.0f KB (x%.3f)" % (k, d.get("FETCH_SIZE_KB", 0), d.get("FETCH_SIZE_KB", 0) / MB, d.get("WRITE_SIZE_KB", 0), d.get("WRITE_SIZE_KB", 0) / MB))
json.dump(out, open('/root/repo/gpurun_out/pmc_calib/summary.json', 'w'), indent=1)
PY
rm -rf $OUT/*/*kernel_trace*
