"""Instruction mix of the kernels in a hipcc -S listing:  python tools/isa_stats.py file.s [name-substring]
(MFMA / LDS / vector-memory counts, how the s_waitcnt points are distributed, scratch use)"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
names = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", txt, re.M)]
for i, (pos, name) in enumerate(names):
    if want not in name:
        continue
    end = txt.find("s_endpgm", pos)
    body = txt[pos:end].split("\n")[1:]
    ops = [l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.strip().endswith(":")]
    c = Counter(ops)
    g = lambda f: sum(v for k, v in c.items() if f(k))
    print(name)
    print(f"  instrs {len(ops)}  mfma {g(lambda k: 'mfma' in k)}  ds_read {g(lambda k: k.startswith('ds_read'))}  ds_write {g(lambda k: k.startswith('ds_write'))}  "
          f"global_load {g(lambda k: k.startswith('global_load'))}  global_store {g(lambda k: k.startswith('global_store'))}  waitcnt {c['s_waitcnt']}  "
          f"barrier {c['s_barrier']}  accvgpr {g(lambda k: 'accvgpr' in k)}  scratch {g(lambda k: 'scratch' in k)}  valu {g(lambda k: k.startswith('v_') and 'mfma' not in k and 'accvgpr' not in k)}")
    w = Counter(l.strip() for l in body if "s_waitcnt" in l)
    print("  waits:", ", ".join(f"{k.replace('s_waitcnt ', '')} x{v}" for k, v in w.most_common(10)))
