"""rocprofv3 kernel trace -> how much kernel time overlaps (sum of durations vs union of busy intervals) and the per-kernel mean duration."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # ignore the first `skip` fraction (warm-up) in percent
rows = rows[len(rows) * skip // 100:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
total = sum(e - s for s, e in iv)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f"kernels {len(rows)}  sum of durations {total/1e6:.2f} ms  union busy {union/1e6:.2f} ms  span {span/1e6:.2f} ms  mean concurrency {total/union:.2f}  idle {100*(1-union/span):.1f}%")
acc = defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"][:64]
    acc[k][0] += 1; acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:<64} {v[0]:6d} x {v[1]/v[0]/1e3:8.1f} us")
