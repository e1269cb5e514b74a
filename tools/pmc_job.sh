#!/bin/bash
# rocprofv3 PMC passes over an EAGER run of bench.py (separate passes, no trace domains with --pmc: pool rule).
#   usage: bash tools/pmc_job.sh <tag> [bench args...]       -> gpurun_out/<tag>_pmc_{traffic,mfma}.json, <tag>_pmc_wait.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}; shift || true
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --repeats 1 --no-roofline --no-graph $*"
run() { rm -rf /tmp/pf_$1; timeout 900 rocprofv3 -M --pmc $2 --output-format csv -d /tmp/pf_$1 -o p -- python3 $R/bench.py $ARGS > /tmp/$1.log 2>&1; echo "$1 rc=$?"; }
run f "FETCH_SIZE"
run w "WRITE_SIZE"
run m "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run s "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"
f() { find /tmp/pf_$1 -name "*counter_collection.csv" | head -1; }
python3 $R/tools/pmc_traffic.py $(f f) $(f w) $OUT/${TAG}_pmc_traffic.json
python3 $R/tools/pmc_mfma.py $(f m) $OUT/${TAG}_pmc_mfma.json
python3 $R/tools/pmc_summary.py $(f s) > $OUT/${TAG}_pmc_wait.txt 2>&1
head -30 $OUT/${TAG}_pmc_wait.txt | cut -c1-250
tail -3 /tmp/s.log
