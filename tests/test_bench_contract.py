"""bench.py's driver contract, exercised end to end on the GPU box: the one-line JSON at N = 1 and the N > 1 launch path
(`python bench.py --gpus 2` -> torch.distributed.run -> one rank per process -> captured data-parallel step -> per-kernel probe on EVERY
rank -> rank 0's line -> clean exit).  A one-GPU box cannot host two RCCL ranks, so the N = 2 run is a functional dry run: both ranks on
GPU 0, gradients exchanged over gloo (LY_BENCH_ONE_GPU / LY_BENCH_BACKEND, bench.py) — it found that a rank-0-only probe of the eager step
would have paired its all-reduces with the other ranks' final barrier."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline"}


def _run(extra, env=None, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run(["--steps", "3", "--warmup", "2", "--repeats", "1", "--batch", "8", "--no-cpu-baseline", "--no-secondary"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "bf16" and d["scaling"] == "weak"
    assert d["metric"] == "images/sec (640x640) fwd+bwd" and d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    # HBM is the roof of every bf16-storage kernel — the line's fraction IS the HBM fraction, the other units are side fields
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] == r["hbm_frac"]
    fr = dict(hbm=r["hbm_frac"], mfma=r["mfma_frac"], valu=r["valu_frac"])
    assert r["limited_by"] in ("hbm", "mfma", "valu", "latency") and (r["limited_by"] == "latency") == (max(fr.values()) < 0.25)
    assert r["limited_by"] == "latency" or fr[r["limited_by"]] == max(fr.values())
    assert 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    # the fractions follow from the line's own inputs: algorithmic work of one launch / its mean duration / the unit's peak
    sec = r["ms_per_launch"] * 1e-3
    assert abs(r["algorithmic_bytes_per_launch"] / sec / 8e12 - r["hbm_frac"]) < 2e-3
    assert abs(r["algorithmic_mfma_flops_per_launch"] / sec / 2.5e15 - r["mfma_frac"]) < 2e-3
    assert abs(r["algorithmic_valu_flops_per_launch"] / sec / 157.3e12 - r["valu_frac"]) < 2e-3
    # the named kernel is picked deterministically: the instrumented kernel with the most time per step; the longest single launch beside it
    big = r["largest_single_launch"]
    assert big["ms_per_launch"] >= r["ms_per_launch"] - 1e-9 and r["ms_per_step"] >= big["ms_per_launch"] * big["launches_per_step"] - 1e-3   # both rounded to 4-5 digits
    busy = (r.get("mfma_busy") or {}).get("mfma_busy_frac")
    if busy and d["config"].get("global_batch") == 64:        # the committed PMC pass is the default (bs = 64) workload's
        assert r["mfma_frac"] <= 1.5 * busy + 1e-3, "the MFMA fraction claimed exceeds what the matrix-pipe counter saw"
    assert r["step"]["families"] and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.parametrize("form", ["probe", "serial", "overlapped"])
def test_two_rank_launch_path_dry_run(form):
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "2", "--repeats", "1", "--batch", "4", "--no-cpu-baseline", "--dp-exchange", form],
             env={"LY_BENCH_ONE_GPU": "1", "LY_BENCH_BACKEND": "gloo"})
    pr = d["dp_overlap"].get("probe")
    assert pr and d["dp_overlap"]["rccl_ranks"] == 2, d["dp_overlap"]
    if form == "probe":
        # both exchange forms timed over the real two-rank group (max over ranks), the faster one kept — on every rank the same one
        assert pr["ranks"] == 2 and pr["serial_ms"] > 0 and pr["overlapped_ms"] > 0 and pr["replays"] >= 5, pr
        assert pr["chosen"] == ("serial" if pr["serial_ms"] < pr["overlapped_ms"] else "overlapped") == d["dp_overlap"]["mode"], d["dp_overlap"]
    else:
        assert d["dp_overlap"].get("mode") == form and pr.get("forced"), d["dp_overlap"]
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["world_size"] == 2
    assert "DRY RUN" in d["config"]["parallelism"] and d["grad_buckets"] >= 1
    assert d["roofline"]["step"]["families"]                       # the probe ran (on both ranks) and the job still ended cleanly
    ov = d["dp_overlap"]
    assert d["rccl_ranks"] == 2 and "error" not in ov and ov["graph_a_ms"] > 0 and ov["step_ms"] >= ov["graph_a_ms"]
    if ov["mode"] == "serial":                      # lead-yolo-s' 12.5 MB of gradients: one synchronous all-reduce between the two graphs
        assert "serial exchange" in d["launch_mode"] and ov["bucket_release_pct_of_graph_a"] == []
    else:
        assert "released mid-graph" in d["launch_mode"]
        assert len(ov["bucket_release_pct_of_graph_a"]) >= 1 and all(b["released_at_pct"] > 0 for b in ov["bucket_release_pct_of_graph_a"])
    assert abs(d["value"] - 2 * 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
