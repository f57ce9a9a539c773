import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# find the notify kernels: each ends a prepared call
idx = [i for i, k in enumerate(ks) if k[0].startswith("hl::notify")]
print("notify kernels at", idx[-8:])
for a, b in zip(idx[:-1], idx[1:]):
    seg = ks[a + 1:b + 1]
    if len(seg) < 30: continue
    t0 = seg[0][1]; total = seg[-1][2] - t0
    k1 = [e - s for n, s, e in seg if "fused" in n]; k2 = [e - s for n, s, e in seg if "dw_table" in n]
    gaps = [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
    print("call of %d kernels: total %.1f us; K1 sum %.1f (first three %s) K2 sum %.1f (first three %s) gaps sum %.1f (first four %s)" % (
        len(seg), total / 1e3, sum(k1) / 1e3, [round(x / 1e3, 1) for x in k1[:3]], sum(k2) / 1e3, [round(x / 1e3, 1) for x in k2[:3]], sum(gaps) / 1e3, [round(x / 1e3, 1) for x in gaps[:4]]))
