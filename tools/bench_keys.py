"""stdin: one bench.py JSON line -> the numbers an A/B looks at (step time, forward times, families, the north-star module times)"""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print("step ms", d["ms_per_step"], d.get("repeat_ms_per_step"))
f = d.get("forward") or {}
print("forward ms", {k: v["ms_per_step"] for k, v in f.items()})
r = d.get("roofline") or {}
if "step" in r:
    print("families", {k: (v["launches"], round(v["ms"], 4)) for k, v in r["step"]["families"].items()})
if "pconv_rfcbam_fwd" in r:
    p = r["pconv_rfcbam_fwd"]
    print("pconv_rfcbam_fwd", p["ms"], p["hbm_frac"], "one graph", p.get("one_graph_ms"), p.get("one_graph_hbm_frac"), [m[2] for m in p["us_per_module"]])
if "pconv_rfcbam_fwd_f32" in r:
    p = r["pconv_rfcbam_fwd_f32"]
    print("pconv_rfcbam_fwd_f32", p["ms"], p["hbm_frac"], [m[2] for m in p["us_per_module"]])
