"""Registers and spills of every kernel of the library, from the compiler's own metadata (no GPU needed):

    python tools/kernel_regs.py                 # compiles lead-yolo_amd/csrc/*.hip with -S (8 at a time) and lists kernels that spill
    python tools/kernel_regs.py --all [substr]  # every kernel (optionally: names containing substr)
    python tools/kernel_regs.py file.s ...      # parse existing `hipcc -S --cuda-device-only` listings instead

`vgpr` is the unified count (architectural + accumulation registers, 512 per lane at one wave per SIMD, 256 at two, 168 at three); `spill` is
what went to scratch memory.  Found in round 5: ly_mlpblock_bwd_kernel<80, pass 2> (51 spilled registers) and ly_mlpblock_bwd_dx_kernel<160>
(32) — both fixed by not holding operand sets live beside values that are only needed after the MFMAs."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lead-yolo_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-S", "--cuda-device-only"]


def kernels(listing):
    txt = open(listing).read()
    if "amdhsa.kernels:" not in txt:
        return []
    out = []
    for e in txt[txt.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
        get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", e).group(1))  # noqa: E731
        out.append((re.search(r"\.name:\s+(\S+)", e).group(1), get("vgpr_count"), int(e.split()[0]), get("vgpr_spill_count"),
                    get("private_segment_fixed_size"), get("group_segment_fixed_size")))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all = "--all" in sys.argv
    files = [a for a in args if a.endswith(".s")]
    want = next((a for a in args if not a.endswith(".s")), "")
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="ly_regs_")
        srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))

        def build(src):
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, src, "-o", out], cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return out
        with ThreadPoolExecutor(8) as ex:
            files = [f for f in ex.map(build, srcs) if os.path.exists(f)]
    rows = [(os.path.basename(f).replace(".hip.s", ""), *k) for f in files for k in kernels(f)]
    names = subprocess.run(["c++filt"] + [r[1] for r in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
    n = 0
    for r, name in zip(rows, names):
        if want not in name or not (show_all or r[4] > 0):
            continue
        n += 1
        print(f"{r[0]:20s} vgpr {r[2]:3d} (acc {r[3]:3d})  spill {r[4]:3d}  scratch {r[5]:4d} B  static LDS {r[6]:6d} B  {name[:150]}")
    print(f"{n} of {len(rows)} kernels listed" + ("" if show_all else " (those that spill)"))


if __name__ == "__main__":
    main()
