"""The committed evidence of the round is self-consistent (CPU test: reads files under profiles/ only)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_named_kernel_is_row_0_of_the_committed_step_table():
    """the default line's `roofline.kernel` is the top kernel by time per step — the first kernel row of the round's committed per-step table
    (profiles/r06_train_bf16_step_kernels.txt: one steady-state eager step of the same command)"""
    path = os.path.join(ROOT, "profiles", "r06_train_bf16_bench.json")
    table = os.path.join(ROOT, "profiles", "r06_train_bf16_step_kernels.txt")
    if not (os.path.exists(path) and os.path.exists(table)):
        pytest.skip("round-6 profiles not committed yet")
    d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    rows = [ln for ln in open(table).read().splitlines()[1:] if ln.strip()]
    name = d["roofline"]["kernel"]
    assert rows[0].startswith(name) and d["roofline"]["frac"] == d["roofline"]["hbm_frac"]
    # the whole-process rocprofv3 trace of the same command (kernel-trace --stats, per step) names the same kernel first among the library's
    # kernels (that trace also holds the set-up of the process — ~820 `__amd_rocclr_copyBuffer` device copies of model construction, the EMA
    # deep copy and the state load, divided by the 7 steps like everything else — which the steady-state table above does not)
    trace = os.path.join(ROOT, "profiles", "r06_train_bf16_kernels_per_step.txt")
    if os.path.exists(trace):
        assert [ln for ln in open(trace).read().splitlines()[1:] if ln.strip().startswith("ly_")][0].startswith(name)
