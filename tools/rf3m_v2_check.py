"""A/B of the two forms of ly_rf3m_fwd (csrc/ly_rf3m.hip): form 2 (one wave per SIMD, two tiles per wave) must return the bits of form 1;
per-launch times of both.  GPU box:  python tools/rf3m_v2_check.py [--prof]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import capi, modules as M, ops           # noqa: E402
from oracle import synth                                    # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
lib = capi.lib()
ops.RF3M_MIN_UNITS = 0


def build(ci, co, s, seed):
    m = L.RFCBAMConv(ci, co, 3, s)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), seed)
    m.load_state_dict(st, strict=True)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    return m.to(dev).eval().bfloat16()


def run(m, xd, reps):
    """statistics pass under both forms, then the contraction under both forms on the SAME ca / rfa (those of form 1)"""
    xr, ld = ops.rows(xd)
    n, c, h, w = xr.shape
    s = m.stride
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    P = m._packed(ops.planes_of(xr))
    th, tw = ops.pick_tile_m(ho, wo, s)
    res, per = {}, {}
    with torch.no_grad():
        for form in (1, 2):
            lib.ly_rf3m_set_form(form)
            mm, part = ops.rf3m_stats(xr, ld, n, h, w, c, s, P["wm_stats"], th, tw)
            torch.cuda.synchronize()
            if "-v" in sys.argv:
                print(f"  stats form {form} done", flush=True)
            res[("mm", form)], res[("part", form)] = mm.clone(), part.clone()
        lib.ly_rf3m_set_form(1)
        y = m(xd)                                         # (warms the packed weights; ca / rfa below come from the form-1 statistics)
        ca, rfa = CAP["ca"], CAP["rfa"]
        for form in (1, 2):
            lib.ly_rf3m_set_form(form)
            out = ops.empty_nhwc(n, m.o, ho, wo, xr)
            kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=m.o, s=s, th=th, tw=tw, x=xr, ldx=ld, ca=ca, rfa=rfa, wp=P["wm"], e_scale=P["es"],
                      e_shift=P["eb"], out=out, ldo=m.o)
            ops.rf3m_fwd(**kw)
            torch.cuda.synchronize()
            if "-v" in sys.argv:
                print(f"  fwd form {form} done", flush=True)
            res[("y", form)] = out.clone()
            if reps:
                ops.PROFILE = []
                for _ in range(reps):
                    ops.rf3m_stats(xr, ld, n, h, w, c, s, P["wm_stats"], th, tw)
                    ops.rf3m_fwd(**kw)
                torch.cuda.synchronize()
                for r in ops.PROFILE:
                    per.setdefault((r[0].split("<")[0], form), []).append(r[3].elapsed_time(r[4]) * 1e3)
                ops.PROFILE = None
    return res, {k: sorted(v)[len(v) // 2] for k, v in per.items()}


CAP = {}
_mid = ops.rfcbam_mid


def _mid_capture(*a, **k):
    ca, rfa = _mid(*a, **k)
    CAP["ca"], CAP["rfa"] = ca, rfa
    return ca, rfa


ops.rfcbam_mid = _mid_capture

SHAPES = [(128, 128, 2, (64, 128, 80, 80)), (256, 256, 2, (64, 256, 40, 40)), (128, 128, 2, (3, 128, 80, 80)), (64, 64, 2, (2, 64, 21, 13)),
          (32, 64, 1, (2, 32, 9, 70)), (512, 512, 2, (2, 512, 12, 12)), (256, 512, 2, (5, 256, 33, 47)), (64, 192, 1, (1, 64, 7, 5))]
bad = 0
for ci, co, s, shape in SHAPES:
    m = build(ci, co, s, 11 + ci)
    xd = synth.synth_input(shape, 5 + ci).to(dev).to(BF).contiguous(memory_format=torch.channels_last)
    big = shape[0] >= 32
    res, t = run(m, xd, 7 if big else 0)
    same = torch.equal(res[("y", 1)], res[("y", 2)])
    same_part = torch.equal(res[("part", 1)], res[("part", 2)])
    mm1, mm2 = res[("mm", 1)], res[("mm", 2)]
    mm_ok = torch.equal(mm1[..., 0], mm2[..., 0]) and bool(((mm1[..., 1] - mm2[..., 1]).abs() <= 1e-5 * mm1[..., 1].abs() + 1e-7).all())
    ok = same and same_part and mm_ok
    bad += not ok
    print(f"{'OK ' if ok else 'BAD'} C={ci} O={co} s={s} {shape}: contraction form 2 == form 1: {same}; pooling partials equal: {same_part}; "
          f"[max, mean] map max equal + mean within 1e-5: {mm_ok} (max diff {(mm1 - mm2).abs().max().item():.2e})" +
          (f" | fwd {t.get(('ly_rf3m_fwd_kernel', 1), 0):.1f} -> {t.get(('ly_rf3m_fwd_kernel', 2), 0):.1f} us; stats "
           f"{t.get(('ly_rf3m_stats_kernel', 1), 0):.1f} -> {t.get(('ly_rf3m_stats_kernel', 2), 0):.1f} us" if big else ""), flush=True)
lib.ly_rf3m_set_form(2)
if "--prof" in sys.argv:
    lib.ly_rf3m_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ["prologue", "main reads + generate + relu*ca*rfa A", "-", "wait copies / x (+ x tile store)", "barrier", "issue copies + x prefetch",
             "next reads, main A + relu*ca*rfa B, main B", "tap-8 tiles"]
    for ci, co, s, shape in SHAPES[:2]:
        m = build(ci, co, s, 7)
        xd = torch.randn(shape, device=dev).to(BF).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            m(xd)
            torch.cuda.synchronize()
            lib.ly_rf3m_prof(None, 1)
            for _ in range(3):
                m(xd)
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 10)()
        lib.ly_rf3m_prof(buf, 0)
        tot = sum(buf[i] for i in range(8))
        print(f"PROF C={ci} O={co} {shape}: " + " | ".join(f"{names[i]} {100.0 * buf[i] / max(tot, 1):.1f}%" for i in range(8)) +
              f" | cycles per sampled wave-launch: {tot}", flush=True)
sys.exit(1 if bad else 0)
