"""Summarise rocprofv3 counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:70]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in sorted(acc.items()):
            if not any(s in k for s in ("spmv", "OpCgUpdate", "OpXpby", "fin_alpha", "fin_res")):
                continue
            print(os.path.basename(d), k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
