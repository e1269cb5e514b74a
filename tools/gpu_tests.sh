#!/bin/bash
# GPU parity tests only, all failures reported:  bash tools/gpu_tests.sh <tag> [pytest args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-t}; shift || true
mkdir -p $R/gpurun_out; cd $R
timeout 2000 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider "$@" > gpurun_out/${TAG}_pytest.log 2>&1
echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | cut -c1-220 | tail -70
