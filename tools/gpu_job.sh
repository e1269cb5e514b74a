#!/bin/bash
# One GPU-box job: GPU parity tests, the default bench line, and a rocprofv3 kernel trace of the train step.
#   usage (through gpurun): bash tools/gpu_job.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-job}; shift || true
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} ${K:+-k "$K"} > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -5 $OUT/${TAG}_pytest.log
timeout 900 python bench.py --layers "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_kernels.txt
echo "bench rc=$?"
tail -c 3000 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_k
timeout 600 rocprofv3 -M --kernel-trace --stats --output-format csv -d /tmp/pf_k -o k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-roofline "$@" > /tmp/k.log 2>&1
cp $(find /tmp/pf_k -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_train_kernel_stats.csv 2>/dev/null
python3 $R/tools/prof_summary.py $(find /tmp/pf_k -name "*kernel_trace.csv" | head -1) 7 > $OUT/${TAG}_train_kernels_per_step.txt 2>&1
head -30 $OUT/${TAG}_train_kernels_per_step.txt
