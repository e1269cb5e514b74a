"""A/B check of the lane = channel RFCBAMConv k=3 kernels (csrc/ly_rf3c.hip) against the first-generation kernels and the
oracle, plus per-module timings (hipGraph replay of the eval forward).  Run on the GPU box:  python tools/rf3c_check.py [--time]"""
import copy
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import modules as M, ops                 # noqa: E402
from oracle import functional as OF, synth                  # noqa: E402

dev = torch.device("cuda:0")


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


def check(ci, co, s, shape, dtype):
    m, st = build(ci, co, s, 1234 + ci + shape[2])
    x = synth.synth_input(shape, 99 + ci)
    with torch.no_grad():
        want = OF.rfcbam(copy.deepcopy(st), "", x, 3, s, False)
    m = m.to(dev).eval()
    if dtype == torch.bfloat16:
        m = m.bfloat16()
    xd = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        M.RF3C = False
        y_old = m(xd).float().cpu()
        M.RF3C = True
        y_new = m(xd).float().cpu()
    r_old, r_new, r_on = rel(y_old, want), rel(y_new, want), rel(y_new, y_old)
    ok = r_new[0] < (2e-5 if dtype == torch.float32 else 2e-2) * 5
    print(f"{'OK ' if ok else 'BAD'} C={ci} O={co} s={s} {shape} {str(dtype)[6:]}: old-vs-oracle rel {r_old[0]:.2e} max {r_old[1]:.2e} | new-vs-oracle rel {r_new[0]:.2e} "
          f"max {r_new[1]:.2e} | new-vs-old rel {r_on[0]:.2e} max {r_on[1]:.2e}", flush=True)
    # intermediates: statistics map and pooling partials
    xr, ld = ops.rows(xd)
    n, c, h, w = xr.shape
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    P = m._packed(ops.planes_of(xr))
    th, tw = ops.pick_tile(ho, wo)
    mm_old = ops.rfcbam_stats(xr, ld, n, h, w, c, 3, s, wg=P["wq_stats"], th=th, tw=tw)
    part_old = ops.colsum(xr, ld, n, h * w, c).sum(1)
    th2, tw2 = ops.pick_tile_c(ho, wo, s)
    mm_new, part_new = ops.rf3c_stats(xr, ld, n, h, w, c, s, P["wq_c"], th2, tw2)
    rm, rp = rel(mm_new, mm_old), rel(part_new.sum(1), part_old)
    print(f"      tile {th2}x{tw2}: mm new-vs-old rel {rm[0]:.2e} max {rm[1]:.2e} | gap rel {rp[0]:.2e} max {rp[1]:.2e}", flush=True)
    return ok and rm[0] < 1e-4 and rp[0] < 1e-4


def timeit(ci, co, s, shape, dtype, reps=30):
    m, _ = build(ci, co, s, 7)
    m = m.to(dev).eval()
    if dtype == torch.bfloat16:
        m = m.bfloat16()
    xd = torch.randn(shape, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    res = {}
    for flag in (False, True):
        M.RF3C = flag
        with torch.no_grad():
            for _ in range(3):
                m(xd)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                m(xd)
                with torch.cuda.graph(g, stream=st):
                    y = m(xd)
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res[flag] = e0.elapsed_time(e1) / reps * 1e3
    M.RF3C = True
    print(f"TIME C={ci} O={co} s={s} {shape} {str(dtype)[6:]}: old {res[False]:.1f} us  new {res[True]:.1f} us", flush=True)
    # per-kernel: the two new launches alone
    xr, ld = ops.rows(xd)
    n, c, h, w = xr.shape
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    P = m._packed(ops.planes_of(xr))
    th2, tw2 = ops.pick_tile_c(ho, wo, s)
    mm, part = ops.rf3c_stats(xr, ld, n, h, w, c, s, P["wq_c"], th2, tw2)
    wa, wb = m.se.fc[0].weight.detach().float().contiguous(), m.se.fc[2].weight.detach().float().contiguous()
    ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, m.se.ratio, mm, P["w18"])
    out = ops.empty_nhwc(n, co, ho, wo, xr)
    kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=co, s=s, th=th2, tw=tw2, x=xr, ldx=ld, wq=P["wq_c"], ca=ca, rfa=rfa, wp=P["wp_c"], ldo=co)

    def t(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_st = t(lambda: ops.rf3c_stats(xr, ld, n, h, w, c, s, P["wq_c"], th2, tw2))
    t_fw = t(lambda: ops.rf3c_fwd(out=out, e_scale=P["es"], e_shift=P["eb"], **kw))
    print(f"      rf3c_stats {t_st:.1f} us   rf3c_fwd {t_fw:.1f} us   (eager, back to back, tile {th2}x{tw2})", flush=True)


if __name__ == "__main__":
    cases = [(128, 128, 2, (2, 128, 80, 80)), (256, 256, 2, (2, 256, 40, 40)), (64, 64, 2, (1, 64, 21, 13)), (32, 48, 1, (2, 32, 9, 70)),
             (512, 512, 2, (1, 512, 12, 12)), (64, 32, 1, (1, 64, 5, 6)), (96, 200, 2, (3, 96, 33, 18))]
    good = True
    for dt in (torch.float32, torch.bfloat16):
        for ci, co, s, shape in cases:
            try:
                good &= check(ci, co, s, shape, dt)
            except Exception as e:                      # keep going: one report per case
                good = False
                print(f"EXC C={ci} O={co} s={s} {shape} {dt}: {type(e).__name__}: {e}", flush=True)
    print("ALL OK" if good else "FAILURES", flush=True)
    if "--time" in sys.argv:
        for dt in (torch.bfloat16, torch.float32):
            timeit(128, 128, 2, (64, 128, 80, 80), dt)
            timeit(256, 256, 2, (64, 256, 40, 40), dt)
