#!/bin/bash
# round 6: one GPU-box job = a list of commands with their logs under gpurun_out/<tag>_<n>.log
#   usage (through gpurun): bash tools/r6_job.sh <tag> "<cmd 1>" "<cmd 2>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-job}; shift || true
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
i=0
for c in "$@"; do
  i=$((i + 1))
  echo "=== [$i] $c" > $OUT/${TAG}_$i.log
  timeout 1200 bash -c "$c" >> $OUT/${TAG}_$i.log 2>&1
  echo "rc=$?" >> $OUT/${TAG}_$i.log
  tail -n 12 $OUT/${TAG}_$i.log
done
