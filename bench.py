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


BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16; fp32-grade products take 3 bf16 MFMAs (csrc/ly_tile.cuh)


def roofline_probe(model, x, iters=10):
    """Times every C-ABI launch of the step with HIP events on the launch stream (ops.PROFILE hooks), groups
    them by the kernel name rocprofv3 prints, and returns the per-kernel table.  Algorithmic bytes:
    input once + output once + parameters once; algorithmic flops: 2*MAC."""
    from lead_yolo_amd import ops
    with torch.no_grad():
        model(x)
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(iters):
            model(x)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
    table = {}
    for name, flops, nbytes, e0, e1 in recs:
        t = table.setdefault(name, dict(kernel=name, calls=0, ms=0.0, flops=0.0, bytes=0.0))
        t["calls"] += 1
        t["ms"] += e0.elapsed_time(e1)
        t["flops"] += flops
        t["bytes"] += nbytes
    rows = []
    for t in table.values():
        c = t["calls"]
        rows.append(dict(kernel=t["kernel"], calls_per_step=c / iters, ms_per_launch=t["ms"] / c, ms_per_step=t["ms"] / iters,
                         flops=t["flops"] / c, bytes=t["bytes"] / c))
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/*_pmc_traffic.json, produced
    by tools/pmc_traffic.py: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if kernel in d:
            return d[kernel]
    return None


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


def cpu_baseline(scale, size, budget_s=15.0):
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
            if time.perf_counter() - t0 > budget_s or n >= 2000:
                break
        dt = time.perf_counter() - t0
    return dict(value=round(b * n / dt, 2), unit="images/sec", cores=cores, kind="port",
                sample=f"{n} eval forwards of lead-yolo-{scale} bs={b} {size}x{size} fp32 (oracle/functional.py, torch {torch.__version__} CPU, "
                       f"{cores} threads) in {dt:.1f}s")


def train_bench(args, rank, local_rank, world, dist, device, barrier):
    """BASELINE.json configs[2]-[3] shape (lead-yolo-s, train mode, forward + loss + backward + SGD step), in fp32 with
    bf16x3 products (the bf16-autocast variant of those configs is not built yet).  One process per GPU, per-GPU batch
    fixed (weak scaling), gradients averaged by ddp.GradReducer over RCCL, overlapped with backward."""
    import lead_yolo_amd as L
    torch.manual_seed(0)
    model = L.Model(L.load_cfg(scale=args.scale)).to(device).train()
    opt = L.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4 * args.batch * world / 64)
    loss_fn = L.ComputeLoss(model)
    reducer = L.GradReducer(list(model.parameters())).attach() if world > 1 else None
    g = torch.Generator().manual_seed(rank)
    imgs = torch.randint(0, 256, (args.batch, 3, args.size, args.size), dtype=torch.uint8, generator=g).to(device)
    nb = 7 * args.batch                                                        # ~7 boxes per image (COCO mean)
    g1 = torch.Generator().manual_seed(1 + rank)
    tg = torch.cat((torch.sort(torch.randint(0, args.batch, (nb, 1), generator=g1).float(), 0)[0], torch.zeros(nb, 1),
                    torch.rand(nb, 2, generator=g1) * 0.8 + 0.1, torch.rand(nb, 2, generator=g1) * 0.2 + 0.02), 1).to(device)
    for _ in range(args.warmup):
        L.train_step(model, loss_fn, opt, imgs, tg, reducer=reducer, world_size=world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = L.train_step(model, loss_fn, opt, imgs, tg, reducer=reducer, world_size=world)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec (640x640) fwd+bwd+SGD", "value": round(world * args.batch * args.steps / dt, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"lead-yolo-{args.scale} bs={args.batch}/gpu 3x{args.size}x{args.size} train step: forward, ComputeLoss, "
                                   "HIP backward, clip 10, SGD-nesterov 3 groups (BASELINE.json configs[2] shape, fp32 instead of bf16 autocast)",
                       "global_batch": world * args.batch, "parallelism": f"dp{world} (bucketed gradient all-reduce over RCCL, overlapped)"},
            "final_loss": round(float(loss), 4), "roofline": None, "cpu_baseline": None,
            "note": "secondary line (bench.py --train); the headline metric is the default eval-forward run"}))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


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
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of the captured hipGraph (serving mode)")
    ap.add_argument("--train", action="store_true",
                    help="secondary line: full optimisation step (forward, loss, backward, clip, SGD-nesterov; gradient all-reduce when N>1) "
                         "instead of the headline eval forward")
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

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    if args.train:
        return train_bench(args, rank, local_rank, world, dist, device, barrier)

    model = build_model(args.scale, device)
    x = synth_batch(args.batch, args.size, rank, device)

    # serving mode: the forward captured once into a hipGraph and replayed (identical kernels, no host work per step);
    # the replayed outputs are checked against an eager forward, any capture problem falls back to eager launches
    step, launch = (lambda: model(x)), "eager"
    if not args.no_graph:
        try:
            import lead_yolo_amd as L
            g = L.GraphedForward(model, x)
            with torch.no_grad():
                ref = model(x)
            got = g()
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip([got[0], *got[1]], [ref[0], *ref[1]])):
                raise RuntimeError("graph replay differs from the eager forward")
            step, launch = g, "hipGraph replay"
        except Exception as e:                                  # noqa: BLE001
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
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
        print(json.dumps({"value": round(value, 2), "ms_per_step": round(ms_per_step, 4), "launch": launch, "note": "probe skipped"}))
    elif rank == 0:
        rows = roofline_probe(model, x)
        if args.layers:
            for r in rows:
                print(f"  {r['kernel']:<46} {r['calls_per_step']:5.1f}/step {r['ms_per_launch'] * 1e3:8.1f} us  {r['ms_per_step'] * 1e3:8.1f} us/step  "
                      f"{r['bytes'] / r['ms_per_launch'] / 1e6:8.1f} GB/s  {r['flops'] / r['ms_per_launch'] / 1e9:8.2f} TFLOP/s", file=sys.stderr)
        dom = rows[0]                                   # dominant kernel = largest share of the step
        gbs = dom["bytes"] / dom["ms_per_launch"] / 1e6
        tfs = dom["flops"] / dom["ms_per_launch"] / 1e9
        hbm_frac = gbs / HBM_PEAK_GBS
        mfma_peak = BF16_MFMA_PEAK_TFLOPS / 3.0
        mfma_frac = tfs / mfma_peak
        if mfma_frac >= hbm_frac:
            roof = dict(bound="mfma", achieved=round(tfs, 2), peak=round(mfma_peak, 1), unit="TFLOP/s", frac=round(mfma_frac, 4),
                        peak_note="dense bf16 2500 TFLOP/s / 3: fp32-grade products = 3 bf16 MFMAs (exact-f32 MFMA peak would be 157.3)")
        else:
            roof = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_frac, 4))
        tr = pmc_traffic(dom["kernel"])
        roof["traffic"] = tr
        roof["kernel"] = dom["kernel"]
        roof["launches_per_step"] = round(dom["calls_per_step"], 2)
        roof["ms_per_launch"] = round(dom["ms_per_launch"], 5)
        roof["algorithmic_bytes_per_launch"] = round(dom["bytes"])
        roof["algorithmic_flops_per_launch"] = round(dom["flops"])
        star = [r for r in rows if r["kernel"].startswith(("ly_mlpblock", "ly_rfcbam"))]
        sb, sm = sum(r["bytes"] * r["calls_per_step"] for r in star), sum(r["ms_per_step"] for r in star)
        roof["pconv_rfcbam_fwd"] = dict(ms=round(sm, 4), hbm_frac=round(sb / sm / 1e6 / HBM_PEAK_GBS, 4),
                                        note="mlpblock + rfcbam stats/main kernels; SE, rfa map and the k=1 GEMM excluded")
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.scale, args.size)
        out = {
            "metric": "images/sec (640x640) forward", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"lead-yolo-{args.scale} bs={args.batch}/gpu 3x{args.size}x{args.size} fp32 eval forward "
                                   "(BASELINE.json configs[1]); random-init weights, perturbed BN stats",
                       "global_batch": world * args.batch, "parallelism": f"dp{world} (independent replicas, no data-path collective)",
                       "launch": launch},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
