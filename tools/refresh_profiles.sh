#!/bin/bash
# Regenerates the rocprofv3 evidence of a round under profiles/ (run on the GPU box through gpurun; profiles/ travels back via
# gpurun_out/, copy from there).   usage: bash tools/refresh_profiles.sh <tag>        e.g. r02
# Produces, for the bf16 train step (the bench line), the fp32 train step and the fp32 / bf16 eval forward:
#   <tag>_<run>_kernels_per_step.txt   rocprofv3 -M --kernel-trace --stats of an EAGER run (per-kernel time per step, names demangled by tools/demangle.py)
#   <tag>_<run>_kernel_stats.csv       rocprofv3's own --stats summary of the same run
#   <tag>_<run>_pmc_traffic.json       FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch, separate --pmc passes
#   <tag>_<run>_pmc_mfma.json          SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) per launch
#   <tag>_<run>_pmc_wait.txt           SQ wait / LDS counters per launch
#   <tag>_<run>_bench.json / _kernels.txt   the bench line (graph replay) and bench.py --layers' live HIP-event table
#   <tag>_train_<dtype>_step_kernels.txt  tools/step_kernels.py: device time per kernel of ONE steady-state eager step
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() {   # name, steps-divisor, bench args...
  local NAME=$1; shift; local ARGS="$*"
  rm -rf /tmp/pf_k
  timeout 900 rocprofv3 -M --kernel-trace --stats --output-format csv -d /tmp/pf_k -o k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-roofline --no-graph --no-sustained --no-cpu-baseline --no-secondary $ARGS > /tmp/k.log 2>&1
  cp $(find /tmp/pf_k -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${NAME}_kernel_stats.csv 2>/dev/null
  python3 $R/tools/prof_summary.py $(find /tmp/pf_k -name "*kernel_trace.csv" | head -1) 7 > $OUT/${TAG}_${NAME}_kernels_per_step.txt 2>&1
  local PA="--steps 2 --warmup 1 --repeats 1 --no-roofline --no-graph --no-sustained --no-cpu-baseline --no-secondary $ARGS"
  for p in "f FETCH_SIZE" "w WRITE_SIZE" "m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "s SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    set -- $p; local k=$1; shift
    rm -rf /tmp/pf_$k
    timeout 900 rocprofv3 -M --pmc $* --output-format csv -d /tmp/pf_$k -o p -- python3 $R/bench.py $PA > /tmp/$k.log 2>&1
  done
  f() { find /tmp/pf_$1 -name "*counter_collection.csv" | head -1; }
  python3 $R/tools/pmc_traffic.py $(f f) $(f w) $OUT/${TAG}_${NAME}_pmc_traffic.json
  python3 $R/tools/pmc_mfma.py $(f m) $OUT/${TAG}_${NAME}_pmc_mfma.json
  python3 $R/tools/pmc_summary.py $(f s) > $OUT/${TAG}_${NAME}_pmc_wait.txt 2>&1
}
one train_bf16 --dtype bf16
one train_f32 --dtype f32
one fwd_f32 --mode forward --dtype f32
one fwd_bf16 --mode forward --dtype bf16
cd $R
# steady-state per-kernel table of ONE eager step (torch profiler, device activity): no model-construction launches in the counts
python3 tools/step_kernels.py bf16 64 400 2>/dev/null | grep -v "^\[W\|Warn" > $OUT/${TAG}_train_bf16_step_kernels.txt
python3 tools/step_kernels.py f32 64 400 2>/dev/null | grep -v "^\[W\|Warn" > $OUT/${TAG}_train_f32_step_kernels.txt
# the bench lines themselves: they read the PMC summaries and name row 0 of the step table — copy those where bench.py looks FIRST
cp $OUT/${TAG}_*_pmc_traffic.json $OUT/${TAG}_*_pmc_mfma.json $OUT/${TAG}_train_*_step_kernels.txt $R/profiles/ 2>/dev/null
python3 bench.py --dtype bf16 --layers > $OUT/${TAG}_train_bf16_bench.json 2> $OUT/${TAG}_train_bf16_kernels.txt
python3 bench.py --dtype f32 --layers > $OUT/${TAG}_train_f32_bench.json 2> $OUT/${TAG}_train_f32_kernels.txt
python3 bench.py --mode forward --dtype f32 --layers > $OUT/${TAG}_fwd_f32_bench.json 2> $OUT/${TAG}_fwd_f32_kernels.txt
python3 bench.py --mode forward --dtype bf16 --layers --no-cpu-baseline > $OUT/${TAG}_fwd_bf16_bench.json 2> $OUT/${TAG}_fwd_bf16_kernels.txt
ls -la $OUT | head -40
