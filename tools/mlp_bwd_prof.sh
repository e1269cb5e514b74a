#!/bin/bash
# PMC passes over tools/mlp_bwd_time.py (fused MLPBlock backward kernels):  bash tools/mlp_bwd_prof.sh <tag> [C ...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-mlpb}; shift || true
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pf_$1; timeout 300 rocprofv3 -M --pmc $2 --output-format csv -d /tmp/pf_$1 -o p -- python3 $R/tools/mlp_bwd_time.py "${@:3}" > /tmp/$1.log 2>&1; echo "$1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "$@"
run b "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "$@"
f() { find /tmp/pf_$1 -name "*counter_collection.csv" | head -1; }
python3 $R/tools/pmc_summary.py $(f a) > $OUT/${TAG}_pmc_a.txt 2>&1
python3 $R/tools/pmc_summary.py $(f b) > $OUT/${TAG}_pmc_b.txt 2>&1
grep -E "kernel|mlpblock_bwd" $OUT/${TAG}_pmc_a.txt | cut -c1-300
grep -E "kernel|mlpblock_bwd" $OUT/${TAG}_pmc_b.txt | cut -c1-300
