"""RFCBAMConv k=3 training step, module level: gradients of the recompute backward (csrc/ly_rf3c_bwd.hip) and of the first-generation
backward against the fp32 oracle's autograd, bf16 storage.   python tools/rf3c_bwd_check.py [--time]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lead_yolo_amd as L                                   # noqa: E402
from lead_yolo_amd import modules as M                      # noqa: E402
from oracle import functional as OF, synth                  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def run(ci, co, shape, flag, st, x, dy, dtype):
    M.RF3C = flag
    m = L.RFCBAMConv(ci, co, 3, 2)
    m.load_state_dict(copy.deepcopy(st), strict=True)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    m = m.to(dev).train()
    xd = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(xd)
    y.backward(dy.to(dev).to(dtype).contiguous(memory_format=torch.channels_last))
    g = {k: v.grad.detach().float().cpu() for k, v in m.named_parameters() if v.grad is not None}
    return y.detach().float().cpu(), xd.grad.detach().float().cpu(), g


def check(ci, co, shape):
    m0 = L.RFCBAMConv(ci, co, 3, 2)
    st = synth.synth_state(synth.shapes_of(m0.state_dict()), 4321 + ci + shape[2])
    x = synth.synth_input(shape, 17 + ci).to(torch.bfloat16).float()
    # oracle (fp32 autograd on the bf16-rounded input)
    so = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k) for k, v in copy.deepcopy(st).items()}
    xo = x.clone().requires_grad_(True)
    yo = OF.rfcbam(so, "", xo, 3, 2, True)
    torch.manual_seed(5)
    dy = torch.randn_like(yo) * (yo.detach().abs() > 1e-4)
    yo.backward(dy)
    res = {}
    for flag in (False, True):
        res[flag] = run(ci, co, shape, flag, st, x, dy, torch.bfloat16)
    print(f"C={ci} O={co} {shape}:  rel-L2 vs fp32 oracle   old | new     (new vs old)")
    print(f"   y      {rel(res[False][0], yo.detach()):.2e} | {rel(res[True][0], yo.detach()):.2e}    ({rel(res[True][0], res[False][0]):.2e})")
    print(f"   dx     {rel(res[False][1], xo.grad):.2e} | {rel(res[True][1], xo.grad):.2e}    ({rel(res[True][1], res[False][1]):.2e})")
    ok = rel(res[True][1], xo.grad) < max(3e-2, 2.0 * rel(res[False][1], xo.grad))
    for k in sorted(res[True][2]):
        go = so[k].grad
        if go is None:
            continue
        a, b = rel(res[False][2][k], go), rel(res[True][2][k], go)
        flagged = "" if b < max(3e-2, 2.0 * a) or go.norm() < 1e-6 else "   <-- BAD"
        ok = ok and not flagged
        print(f"   {k:28s} {a:.2e} | {b:.2e}    ({rel(res[True][2][k], res[False][2][k]):.2e}){flagged}")
    return ok


def timeit(ci, co, shape, reps=10):
    from lead_yolo_amd import ops
    m0 = L.RFCBAMConv(ci, co, 3, 2)
    st = synth.synth_state(synth.shapes_of(m0.state_dict()), 1)
    x = torch.randn(shape)
    for flag in (False, True):
        M.RF3C = flag
        m = L.RFCBAMConv(ci, co, 3, 2)
        m.load_state_dict(copy.deepcopy(st))
        m = m.to(dev).train()
        xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        dy = None
        ts = []
        for it in range(reps + 3):
            y = m(xd)
            if dy is None:
                dy = torch.randn_like(y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y.backward(dy)
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"TIME backward C={ci} O={co} {shape} {'new' if flag else 'old'}: median {ts[len(ts) // 2]:.0f} us  min {ts[0]:.0f} us (eager, incl. host gaps)", flush=True)
        ops.PROFILE = []
        y = m(xd)
        y.backward(dy)
        torch.cuda.synchronize()
        for r in ops.PROFILE:
            if "rf" in r[0]:
                print(f"      {r[0]:60s} {r[3].elapsed_time(r[4]) * 1e3:8.1f} us")
        ops.PROFILE = None
    M.RF3C = True


if __name__ == "__main__":
    good = True
    for ci, co, shape in ([(128, 128, (1, 128, 12, 40))] if "--one" in sys.argv else []) + [(64, 64, (1, 64, 21, 13)), (128, 128, (2, 128, 40, 40)), (256, 256, (2, 256, 20, 20)), (32, 64, (3, 32, 16, 24)), (128, 128, (2, 128, 80, 80))]:
        try:
            good &= check(ci, co, shape)
        except Exception as e:
            import traceback
            traceback.print_exc()
            good = False
    print("ALL OK" if good else "FAILURES", flush=True)
    if "--time" in sys.argv:
        timeit(128, 128, (64, 128, 80, 80))
        timeit(256, 256, (64, 256, 40, 40))
