#!/bin/bash
# usage: ab.sh <cmd...>  — runs the command with the default lib, then with each gpurun_in/lib_*.so swapped in
R=${GRAFT_REPO_ROOT:-/root/repo}; L=$R/lead-yolo_amd/csrc/libleadyolo_hip.so
cp $L /tmp/lib_default.so
echo "== default"; eval "$@"
for f in $R/gpurun_in/lib_*.so; do echo "== $(basename $f)"; cp $f $L; eval "$@"; done
cp /tmp/lib_default.so $L
