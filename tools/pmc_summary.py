"""Dev tool: summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void bssm::", "").replace("bssm::", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("%-28s" % "kernel" + "".join("%18s" % c[-17:] for c in names))
for k, d in sorted(acc.items()):
    if not k.startswith("k_"): continue
    print("%-28s" % k[:28] + "".join("%18.0f" % (sum(d[c]) / max(len(d[c]), 1)) for c in names))
