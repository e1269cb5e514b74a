"""Summarise a rocprofv3 kernel trace of `bench.py --no-roofline`: per-kernel time per step and GPU-busy vs wall."""
import csv
import os
import sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demangle import norm  # noqa: E402

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last `steps` worth of the timed region: find by kernel-count periodicity is overkill; report totals / (steps+warmup)
tot = defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = norm(r["Kernel_Name"])
    tot[k][0] += 1
    tot[k][1] += d
n = steps
busy = sum(v[1] for v in tot.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"kernels={len(rows)} busy={busy/1e3:.2f} ms span={span/1e3:.2f} ms  per-step busy={busy/n/1e3:.3f} ms")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[:100]:<100} {v[0]/n:6.1f}/step {v[1]/v[0]:9.1f} us avg {v[1]/n:9.1f} us/step")
