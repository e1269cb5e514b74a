"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv
import os
import sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demangle import norm  # noqa: E402

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(path)):
    k = norm(r["Kernel_Name"])[:70]
    c = r["Counter_Name"]
    acc[k][c][0] += 1
    acc[k][c][1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(70), *[n[-18:].rjust(19) for n in names])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0]))[1]):
    if "ly_" not in k:
        continue
    print(k.ljust(70), *[f"{acc[k][n][1] / max(acc[k][n][0], 1):19.0f}" for n in names])
