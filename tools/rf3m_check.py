"""A/B check of the MFMA-generate RFCBAMConv k=3 kernels (csrc/ly_rf3m.hip) against the lane = channel kernels and the oracle, plus
per-module timings (hipGraph replay of the eval forward) and per-launch times.  GPU box:  python tools/rf3m_check.py [--time]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import modules as M, ops                 # noqa: E402
from oracle import functional as OF, synth                  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def build(ci, co, s, seed):
    m = L.RFCBAMConv(ci, co, 3, s)
    st = synth.synth_state(synth.shapes_of(m.state_dict()), seed)
    m.load_state_dict(st, strict=True)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    return m, st


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item(), (a - b).abs().max().item()


def check(ci, co, s, shape):
    m, st = build(ci, co, s, 1234 + ci + shape[2])
    x = synth.synth_input(shape, 99 + ci).to(BF).float()
    with torch.no_grad():
        want = OF.rfcbam(copy.deepcopy(st), "", x, 3, s, False)
    m = m.to(dev).eval().bfloat16()
    xd = x.to(dev).to(BF).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        M.RF3M = False
        y_c = m(xd).float().cpu()
        M.RF3M = True
        y_m = m(xd).float().cpu()
    r_c, r_m, r_mc = rel(y_c, want), rel(y_m, want), rel(y_m, y_c)
    ok = r_m[0] < 2e-2
    print(f"{'OK ' if ok else 'BAD'} C={ci} O={co} s={s} {shape}: rf3c-vs-oracle rel {r_c[0]:.2e} max {r_c[1]:.2e} | rf3m-vs-oracle rel {r_m[0]:.2e} max {r_m[1]:.2e} | "
          f"rf3m-vs-rf3c rel {r_mc[0]:.2e}", flush=True)
    xr, ld = ops.rows(xd)
    n, c, h, w = xr.shape
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    P = m._packed(ops.planes_of(xr))
    th2, tw2 = ops.pick_tile_c(ho, wo, s)
    mm_c, part_c = ops.rf3c_stats(xr, ld, n, h, w, c, s, P["wq_c"], th2, tw2)
    th, tw = ops.pick_tile_m(ho, wo, s)
    mm_m, part_m = ops.rf3m_stats(xr, ld, n, h, w, c, s, P["wm_stats"], th, tw)
    rm, rp = rel(mm_m, mm_c), rel(part_m.sum(1), part_c.sum(1))
    print(f"      tile {th}x{tw}: mm rf3m-vs-rf3c rel {rm[0]:.2e} max {rm[1]:.2e} | gap rel {rp[0]:.2e} max {rp[1]:.2e}", flush=True)
    return ok and rm[0] < 1e-2 and rp[0] < 1e-5


def timeit(ci, co, s, shape, reps=30):
    m, _ = build(ci, co, s, 7)
    m = m.to(dev).eval().bfloat16()
    xd = torch.randn(shape, device=dev).to(BF).contiguous(memory_format=torch.channels_last)
    for flag in (False, True):
        M.RF3M = flag
        with torch.no_grad():
            for _ in range(3):
                m(xd)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                with torch.cuda.graph(g):
                    m(xd)
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            tot = e0.elapsed_time(e1) / reps * 1e3
            ops.PROFILE = []
            for _ in range(5):
                m(xd)
            torch.cuda.synchronize()
            per = {}
            for r in ops.PROFILE:
                per.setdefault(r[0], []).append(r[3].elapsed_time(r[4]) * 1e3)
            ops.PROFILE = None
        print(f"TIME C={ci} O={co} {shape} {'rf3m' if flag else 'rf3c'}: module {tot:.1f} us (graph) | " +
              " | ".join(f"{k.split('<')[0]} {sorted(v)[len(v) // 2]:.1f}" for k, v in per.items()), flush=True)


def prof():
    """LY_RM_PROF=1 python tools/rf3m_check.py --prof : share of the contraction kernel's wave time per phase (s_memtime stamps)"""
    import ctypes
    from lead_yolo_amd import capi
    lib = capi.lib()
    lib.ly_rf3m_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ["prologue", "phase 1 (main reads + generate MFMAs)", "x tile store", "wait weight copies", "barrier", "issue copies + x prefetch",
             "phase 2 (reads, relu*ca*rfa, main MFMAs)", "tap-8 tile"]
    for ci, co, s, shape in [(128, 128, 2, (64, 128, 80, 80)), (256, 256, 2, (64, 256, 40, 40))]:
        m, _ = build(ci, co, s, 7)
        m = m.to(dev).eval().bfloat16()
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
        clk = buf[8] / max(buf[9], 1) * 0.1
        buf = list(buf)[:8]
        tot = sum(buf)
        print(f"PROF C={ci} O={co}: " + " | ".join(f"{n} {100.0 * v / tot:.1f}%" for n, v in zip(names, buf)) + f" | in-kernel clock {clk:.2f} GHz | cycles per wave-tile {tot / 3 / (-(-(shape[0] * (shape[2] // 2) * (shape[3] // 2) // 32) // 8 // 16) * (co // 128)):.0f}", flush=True)


if __name__ == "__main__":
    if "--batches" in sys.argv:
        for bs in (4, 8, 16, 32):
            timeit(128, 128, 2, (bs, 128, 80, 80))
            timeit(256, 256, 2, (bs, 256, 40, 40))
        sys.exit(0)
    if "--prof" in sys.argv:
        prof()
        sys.exit(0)
    if "--timeonly" in sys.argv:
        timeit(128, 128, 2, (64, 128, 80, 80))
        timeit(256, 256, 2, (64, 256, 40, 40))
        sys.exit(0)
    good = True
    for args in [(64, 64, 2, (2, 64, 16, 16)), (128, 128, 2, (2, 128, 24, 20)), (256, 256, 2, (1, 256, 14, 18)), (32, 64, 1, (2, 32, 9, 14)),
                 (128, 128, 2, (3, 128, 80, 80)), (256, 256, 2, (2, 256, 40, 40))]:
        good &= check(*args)
    print("ALL OK" if good else "FAILURES", flush=True)
    if "--timeonly" in sys.argv:
        timeit(128, 128, 2, (64, 128, 80, 80))
        timeit(256, 256, 2, (64, 256, 40, 40))
        sys.exit(0)
    if "--time" in sys.argv:
        timeit(128, 128, 2, (64, 128, 80, 80))
        timeit(256, 256, 2, (64, 256, 40, 40))
