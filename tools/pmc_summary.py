"""Summarise rocprofv3 --pmc counter_collection.csv files: one line per (kernel, counter), averaged per dispatch."""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            short = name.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
            acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
    print(path)
    for (k, c), v in sorted(acc.items()):
        print(f"  {k:40s} {c:28s} dispatches={len(v):3d} avg={sum(v) / len(v):.6g}")
