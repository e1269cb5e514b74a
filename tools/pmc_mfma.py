"""rocprofv3 PMC csv (SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE pass) -> {kernel: MFMA-busy figures per launch}.

usage: pmc_mfma.py <counter_collection.csv> <out.json>
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16), summed
over the SIMDs that ran the kernel.  GRBM_GUI_ACTIVE counts busy cycles PER XCD and rocprofv3 reports the sum over the 8 XCDs
(checked: GRBM_GUI_ACTIVE / 8 / 2.4 GHz = the kernel-trace duration + ~5 us of dispatch), so the dispatch lasted GRBM_GUI_ACTIVE / 8
cycles and
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
= the fraction of the chip's matrix-pipe capacity the launch used (1.0 = every SIMD's matrix pipe busy every cycle = the dense peak).
Cross-check with the algorithmic FLOPs of bench.py: flops / (2*16*16*32) MFMAs * 16 cycles (x3 for the bf16x3 fp32-storage path)
must not exceed mfma_busy_cycles (it is ~0.6-0.7 of it: ragged tiles and K padded to the staging chunk are issued as well)."""
import csv
import os
import json
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collections import defaultdict

SIMDS = 256 * 4
XCDS = 8


from demangle import norm  # noqa: E402


acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = norm(r["Kernel_Name"])
    c = r["Counter_Name"]
    acc[k][c][0] += 1
    acc[k][c][1] += float(r["Counter_Value"])
out = {}
for k, cs in acc.items():
    if "ly_" not in k:
        continue
    mean = {c: v[1] / v[0] for c, v in cs.items()}
    busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES"), mean.get("GRBM_GUI_ACTIVE")
    if busy is None or not gui:
        continue
    out[k] = dict(mfma_busy_cycles=round(busy), gpu_active_cycles=round(gui), sq_busy_cycles=round(mean.get("SQ_BUSY_CYCLES", 0)),
                  mfma_busy_frac=round(busy / (gui / XCDS * SIMDS), 4), launches=cs["GRBM_GUI_ACTIVE"][0],
                  note="mean per launch; frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)")
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(f"{len(out)} kernels -> {sys.argv[2]}")
