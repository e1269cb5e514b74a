cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_k
timeout 900 rocprofv3 -M --kernel-trace --stats --output-format csv -d /tmp/pf_k -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-roofline --no-graph --dtype bf16 > /tmp/k.log 2>&1
cp $(find /tmp/pf_k -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/q_kernel_stats.csv
