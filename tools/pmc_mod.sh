#!/bin/bash
# two rocprofv3 --pmc passes (wait / issue counters) + a kernel trace over tools/mod_time.py -> gpurun_out/<tag>_pmc_mod.txt
#   usage: bash tools/pmc_mod.sh <tag> <mod_time args...>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}; shift || true
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_a /tmp/pm_b /tmp/pm_k
timeout 600 rocprofv3 -M --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pm_a -o p -- python3 $R/tools/mod_time.py "$@" > /tmp/a.log 2>&1
timeout 600 rocprofv3 -M --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm_b -o p -- python3 $R/tools/mod_time.py "$@" > /tmp/b.log 2>&1
timeout 600 rocprofv3 -M --kernel-trace --stats --output-format csv -d /tmp/pm_k -o k -- python3 $R/tools/mod_time.py "$@" > /tmp/k.log 2>&1
{
  python3 $R/tools/pmc_summary.py $(find /tmp/pm_a -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $(find /tmp/pm_b -name "*counter_collection.csv" | head -1)
  head -12 $(find /tmp/pm_k -name "*kernel_stats.csv" | head -1)
} > $OUT/${TAG}_pmc_mod.txt 2>&1
cut -c1-260 $OUT/${TAG}_pmc_mod.txt | head -40
