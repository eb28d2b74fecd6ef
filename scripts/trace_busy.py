"""GPU busy time vs wall span from a rocprofv3 kernel trace CSV (development aid):
    python scripts/trace_busy.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"kernels {len(rows)}  span {(t1-t0)/1e6:.2f} ms  busy {busy/1e6:.2f} ms  ({100*busy/(t1-t0):.0f} %)")
gaps = collections.Counter()
prev_end, prev_name = None, None
gap_by = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if prev_end is not None:
        g = s - prev_end
        key = (prev_name[:28], r["Kernel_Name"].split("(")[0].replace("void ", "")[:28])
        gap_by[key][0] += 1; gap_by[key][1] += max(g, 0)
    prev_end, prev_name = int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")
for k, (c, t) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"gap {k[0]:28s} -> {k[1]:28s} n={c:6d} mean {t/c/1e3:7.2f} us total {t/1e6:7.2f} ms")
dur = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    dur[k][0] += 1; dur[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"kern {k:40s} n={c:6d} mean {t/c/1e3:7.2f} us total {t/1e6:7.2f} ms")
