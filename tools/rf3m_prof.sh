#!/bin/bash
# rocprofv3 PMC passes of the MFMA-generate RFCBAMConv k=3 forward kernels.   usage: bash tools/rf3m_prof.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rf3m}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pf_$1; timeout 300 rocprofv3 -M --pmc $2 --output-format csv -d /tmp/pf_$1 -o p -- python3 $R/tools/rf3m_time.py 3 > /tmp/$1.log 2>&1; echo "$1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
run b "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
run c "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16"
f() { find /tmp/pf_$1 -name "*counter_collection.csv" | head -1; }
for p in a b c; do python3 $R/tools/pmc_summary.py $(f $p) > $OUT/${TAG}_pmc_$p.txt 2>&1; grep -E "kernel|rf3" $OUT/${TAG}_pmc_$p.txt | cut -c1-250; done
tail -2 /tmp/a.log /tmp/b.log /tmp/c.log
rm -rf /tmp/pf_k; timeout 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d /tmp/pf_k -o k -- python3 $R/tools/rf3m_time.py 5 > /tmp/k.log 2>&1
grep -E "rf3|Name" $(find /tmp/pf_k -name "*kernel_stats.csv" | head -1) | cut -c1-200
