"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel (name prefix + grid)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].replace("void ", "")[:44]
    if filt not in k:
        continue
    agg[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
