"""Average rocprofv3 counter_collection.csv per kernel (our kernels only)."""
import collections
import csv
import sys

try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:  # noqa: BLE001
    print("no pmc csv:", e)
    raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"][:36]
    if not (k.startswith("void k_") or k.startswith("k_")):
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
