#!/bin/bash
# the GPU parity suite of one gpurun call: summary line + failures in gpurun_out/<tag>_pytest.log; exit code = pytest's
#   usage: bash tools/gpu_tests.sh <tag> [pytest args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-t}; shift || true
mkdir -p $R/gpurun_out
cd $R
timeout 2000 python -m pytest tests -m gpu -q -x "$@" > $R/gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
grep -E "passed|failed|error" $R/gpurun_out/${TAG}_pytest.log | tail -3
grep -E "^(FAILED|ERROR)|SSDD after" $R/gpurun_out/${TAG}_pytest.log | head -20
echo "pytest rc=$rc"
exit $rc
