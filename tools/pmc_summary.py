"""Aggregate a rocprofv3 --pmc ... --output-format csv run: mean counter value per dispatch, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

src = sys.argv[1]
files = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(cs.values())))
    print(f"{k}  (dispatches {n})")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v) / len(v):16.1f}")
