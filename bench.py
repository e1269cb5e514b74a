"""bench.py — LEAD-YOLO hot path on MI355X.  Contract: see the task statement / DESIGN.md §Measurement.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--scale s] [--size 640]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one forward pass of the detector (BASELINE.json configs[1]: lead-yolo-s, bs=32,
3x640x640, fp32, eval / BN-folded) over one synthetic batch already resident in HBM.  Rank 0 prints
ONE JSON line.  `roofline` is measured live with HIP events on the launch stream for the dominant
kernel of the step; `cpu_baseline` times the oracle (oracle/functional.py, the parity-checked CPU
restatement of the reference) on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32, exact fp32


def build_model(scale, device, seed=0):
    import lead_yolo_amd as L
    torch.manual_seed(seed)
    m = L.Model(L.load_cfg(scale=scale))
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():                      # non-trivial BN statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    return m.to(device).eval()


def synth_batch(b, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (b, 3, size, size), generator=g, dtype=torch.uint8)
    return (u8.float() / 255).to(device)


def time_kernel(fn, iters=20, warm=3):
    """average duration (ms) of one launch, HIP events on the current (= launch) stream"""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def roofline_probe(model, x):
    """Times the individual ★ kernels on this step's real tensors and returns the roofline object of
    the dominant one.  Algorithmic bytes: input once + output once + parameters once;
    algorithmic flops: 2*MAC of the contractions."""
    import lead_yolo_amd as L
    feats = {}
    hooks = []
    for mod in model.model:
        hooks.append(mod.register_forward_hook(lambda m_, inp, out, i=mod.i: feats.__setitem__(i, (inp, out))))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    rows = []
    for mod in model.model:
        blocks = list(mod) if isinstance(mod, torch.nn.Sequential) else [mod]
        inp, out = feats[mod.i]
        xin = inp[0]
        if isinstance(xin, (list, tuple)) or isinstance(xin, L.Lazy) or not isinstance(out, torch.Tensor):
            continue
        if not isinstance(blocks[0], (L.BasicStage, L.RFCBAMConv, L.C3_CA, L.PatchEmbed_FasterNet, L.PatchMerging_FasterNet)):
            continue
        with torch.no_grad():
            ms = time_kernel(lambda: mod(xin))
        nparam = sum(p.numel() for p in mod.parameters())
        per = len(blocks)
        bytes_ = 4.0 * (xin.numel() + out.numel()) * (per if isinstance(blocks[0], L.BasicStage) else 1) + 4.0 * nparam
        rows.append(dict(layer=mod.i, kind=type(blocks[0]).__name__, ms=ms, bytes=bytes_, flops=_flops(blocks, xin, out)))
    return rows


def _flops(blocks, xin, out):
    import lead_yolo_amd as L
    n, c, h, w = xin.shape
    b0 = blocks[0]
    px_out = out.shape[0] * out.shape[2] * out.shape[3]
    if isinstance(b0, L.BasicStage):
        cq = c // 4
        return len(blocks) * 2.0 * px_out * (9 * cq * cq + 4 * c * c)
    if isinstance(b0, (L.PatchEmbed_FasterNet, L.PatchMerging_FasterNet)):
        return 2.0 * px_out * b0.k * b0.k * c * out.shape[1]
    if isinstance(b0, L.RFCBAMConv):
        k = b0.kernel_size
        return 2.0 * px_out * (k * k * c * out.shape[1] + k * k * k * k * c + 18 * k * k) + 2.0 * n * (2 * 16 * c)
    if isinstance(b0, L.C3_CA):
        c_ = b0.c_
        per_px = c * 2 * c_ + len(b0.m) * (c_ * c_ + 9 * c_ * c_) + 2 * c_ * b0.c2
        return 2.0 * px_out * per_px
    return 0.0


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(scale, size, budget_s=20.0):
    """oracle (CPU restatement, parity-pinned) on the host cores, bounded sample of the same workload"""
    import copy
    from oracle import functional as OF
    import lead_yolo_amd as L
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = build_model(scale, "cpu")
    st = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = L.load_cfg(scale=scale)
    b = 4
    x = synth_batch(b, size, 0, "cpu")
    with torch.no_grad():
        OF.model_forward(copy.deepcopy(st), cfg, x, m.stride, training=False)      # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            OF.model_forward(st, cfg, x, m.stride, training=False)
            n += 1
            if time.perf_counter() - t0 > budget_s or n >= 50:
                break
        dt = time.perf_counter() - t0
    return dict(value=round(b * n / dt, 2), unit="images/sec", cores=cores, kind="port",
                sample=f"{n} eval forwards of lead-yolo-{scale} bs={b} {size}x{size} fp32 (oracle/functional.py, torch {torch.__version__} CPU, "
                       f"{cores} threads) in {dt:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--scale", default="s")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print the per-layer kernel table to stderr")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel probe (for whole-step rocprof runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    model = build_model(args.scale, device)
    x = synth_batch(args.batch, args.size, rank, device)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            model(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(x)
        barrier()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    roof = None
    cpu = None
    if rank == 0 and args.no_roofline:
        print(json.dumps({"value": round(value, 2), "ms_per_step": round(ms_per_step, 4), "note": "probe skipped"}))
    elif rank == 0:
        rows = roofline_probe(model, x)
        if args.layers:
            for r in rows:
                print(f"  layer {r['layer']:>2} {r['kind']:<24} {r['ms']:8.3f} ms  {r['bytes'] / r['ms'] / 1e6:8.1f} GB/s  "
                      f"{r['flops'] / r['ms'] / 1e9:8.2f} TFLOP/s", file=sys.stderr)
        dom = max(rows, key=lambda r: r["ms"])
        hbm_frac = dom["bytes"] / dom["ms"] / 1e6 / HBM_PEAK_GBS
        mfma_frac = dom["flops"] / dom["ms"] / 1e9 / F32_MFMA_PEAK_TFLOPS
        if mfma_frac >= hbm_frac:
            roof = dict(bound="mfma", achieved=round(dom["flops"] / dom["ms"] / 1e9, 2), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(mfma_frac, 4), traffic=None)
        else:
            roof = dict(bound="hbm", achieved=round(dom["bytes"] / dom["ms"] / 1e6, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(hbm_frac, 4), traffic=None)
        roof["kernel"] = f"layer {dom['layer']} {dom['kind']}"
        roof["ms_per_launch"] = round(dom["ms"], 4)
        star = [r for r in rows if r["kind"] in ("BasicStage", "RFCBAMConv")]
        roof["pconv_rfcbam_fwd"] = dict(ms=round(sum(r["ms"] for r in star), 4), algorithmic_GB=round(sum(r["bytes"] for r in star) / 1e9, 4),
                                        hbm_frac=round(sum(r["bytes"] for r in star) / sum(r["ms"] for r in star) / 1e6 / HBM_PEAK_GBS, 4))
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.scale, args.size)
        out = {
            "metric": "images/sec (640x640) forward", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"lead-yolo-{args.scale} bs={args.batch}/gpu 3x{args.size}x{args.size} fp32 eval forward "
                                   "(BASELINE.json configs[1]); random-init weights, perturbed BN stats",
                       "global_batch": world * args.batch, "parallelism": f"dp{world} (independent replicas, no data-path collective)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
