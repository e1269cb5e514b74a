import ctypes, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lead_yolo_amd import capi
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"librb_{sys.argv[2]}.so"))
    lib.ly_rf3c_bwd.restype = ctypes.c_int
    lib.ly_rf3c_bwd.argtypes = [ctypes.POINTER(capi.LyRf3cBwdParams), ctypes.c_int, ctypes.c_void_p]
    lib.ly_last_error.restype = ctypes.c_char_p
    n, C, O, H, W, th, tw = [int(v) for v in sys.argv[3:10]]
    passes = [int(v) for v in sys.argv[10].split(",")]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    mo = n * ho * wo
    bf = torch.bfloat16
    x = torch.randn(n, H, W, C, device=dev).to(bf)
    du = torch.randn(mo, O, device=dev).to(bf)
    wq = torch.randn(C * 100, device=dev) * 0.3
    wct = (torch.randn((9 * C // 16) * (O // 32) * 64 * 8, device=dev) * 0.1).to(bf).view(torch.int16)
    ca = torch.rand(n, C, device=dev)
    rfa = torch.rand(n, 3 * ho, 3 * wo, device=dev)
    mm = torch.rand(n, 3 * ho, 3 * wo, 2, device=dev)
    d_mm = torch.randn(n, 3 * ho, 3 * wo, 2, device=dev)
    coef = torch.randn(3, 9 * C, device=dev)
    d_rfa_part = torch.empty(C // 32, n * 9 * ho * wo, device=dev)
    d_ca = torch.empty(n, C, device=dev)
    sums = torch.empty(n, 18 * C, device=dev)
    dwg = torch.empty(n, C * 81, device=dev)
    dx = torch.empty(n, H, W, C, device=dev, dtype=bf)
    dgap = torch.randn(n, C, device=dev)
    dwc = torch.zeros(max(O * 9 * C, 8192), device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    P = capi.LyRf3cBwdParams(n, H, W, C, ho, wo, O, 2, th, tw, p(x), C, p(du), O, p(wq), p(wct), p(ca), p(rfa), p(mm), p(d_mm), p(coef),
                             p(d_rfa_part), p(d_ca), p(sums), p(dwg), p(dx), C, p(dgap), 1.0 / (H * W), p(dwc), 1, 1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    import time
    for it in range(3):
        for ps in passes:
            torch.cuda.synchronize(); t0 = time.time()
            rc = lib.ly_rf3c_bwd(ctypes.byref(P), ps, st)
            if rc:
                print("rc", rc, lib.ly_last_error().decode()); sys.exit(3)
            torch.cuda.synchronize(); print("host us", (time.time() - t0) * 1e6)
    torch.cuda.synchronize()
    nb = n * (C // 32)
    tt_ = dwc.view(-1)[:nb * 8].view(nb, 8).double().mean(0)
    d2 = dwc.view(-1)[4096 * 8:4096 * 8 + nb * 4].view(nb, 4).cpu()
    cu = set((int(a), int(b)) for a, b in zip(d2[:, 0].tolist(), d2[:, 1].tolist()))
    print("blocks", nb, "distinct (hw_id,xcc):", len(cu), "start spread (10ns ticks):", float(d2[:, 2].max() - d2[:, 2].min()), "block life realtime ticks mean/max:", float(d2[:, 3].mean()), float(d2[:, 3].max()))
    names = ["wait B1", "commit+issue", "wait B2", "dcd MFMA", "wait B3", "VALU", "dx out", "reduce/init"]
    print("OK", " | ".join(f"{nm} {v:.0f}" for nm, v in zip(names, tt_.tolist())), "ticks(100MHz) per launch-avg of 3")
    sys.exit(0)
here = os.path.abspath(__file__)
shapes = [(64, 128, 128, 80, 80, 8, 8), (64, 256, 256, 40, 40, 3, 20)]
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["v0"]
PASS = sys.argv[2] if len(sys.argv) > 2 else "2"
for v in variants:
    for sh in shapes:
        r = subprocess.run([sys.executable, here, "child", v] + [str(a) for a in sh] + [PASS], capture_output=True, text=True)
        out = (r.stdout.strip().splitlines() or [""])[-1]
        flt = "FAULT" if "fault" in (r.stderr + r.stdout).lower() else ""
        print(f"{v:8s} {sh}: rc={r.returncode} {out} {flt}", flush=True)
