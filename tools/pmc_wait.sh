#!/bin/bash
# ONE rocprofv3 --pmc pass (wait / LDS counters) over an eager run of bench.py -> gpurun_out/<tag>_pmc_wait.txt
#   usage: bash tools/pmc_wait.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}; shift || true
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --repeats 1 --no-roofline --no-graph --no-sustained --no-cpu-baseline --no-secondary $*"
rm -rf /tmp/pf_s
timeout 900 rocprofv3 -M --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pf_s -o p -- python3 $R/bench.py $ARGS > /tmp/s.log 2>&1
echo "pmc rc=$?"
python3 $R/tools/pmc_summary.py $(find /tmp/pf_s -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_wait.txt 2>&1
head -40 $OUT/${TAG}_pmc_wait.txt | cut -c1-220
