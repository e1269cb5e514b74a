#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/<tag>_* (run on the GPU box through gpurun; copy the files you keep to profiles/).
#   usage: tools/refresh_profiles.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_k /tmp/pf_f /tmp/pf_w
# 1. per-kernel durations (eager launches: one kernel at a time, comparable with the live HIP-event numbers of bench.py)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_k -o k -- python3 $R/bench.py --steps 20 --warmup 3 --no-roofline --no-cpu-baseline --no-graph > /tmp/k.log 2>&1
cp $(find /tmp/pf_k -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $R/tools/prof_summary.py $(find /tmp/pf_k -name "*kernel_trace.csv" | head -1) 23 > $OUT/${TAG}_kernels_per_step.txt
# 2. HBM traffic per launch: separate FETCH_SIZE and WRITE_SIZE passes (no trace domains together with --pmc)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_f -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-graph > /tmp/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pf_w -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-graph > /tmp/w.log 2>&1
python3 $R/tools/pmc_traffic.py $(find /tmp/pf_f -name "*counter_collection.csv" | head -1) $(find /tmp/pf_w -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json
# 3. the bench line (graph replay) and the per-kernel table
cd $R
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json 2>/dev/null
python3 bench.py --layers > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_kernels.txt
tail -1 $OUT/${TAG}_bench.json | cut -c1-600
