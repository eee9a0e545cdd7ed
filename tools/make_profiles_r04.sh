#!/bin/bash
# Round-4 evidence on the GPU box (outputs under gpurun_out/r4prof/, summaries copied to profiles/r04/ afterwards):
#   1. the driver's command as the driver runs it, and the default (200-step) command;
#   2. the rocprofv3 kernel-trace summary of the driver's command (per kernel and per grid: one process, four workloads);
#   3. the two PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass, nothing else traced) for the headline workload
#      and for EllStable in the mirrored layout (new kernels this round).
# Usage: tools/make_profiles_r04.sh [stage ...]   stages: bench stats pmc stable (default: all)
set -e
# (rocprofv3 7.2 segfaults in the traced process' exit handlers AFTER it has written its CSVs and the bench line is out: tolerated)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4prof
mkdir -p $O
export TMPDIR=/tmp
STAGES=${*:-bench stats pmc stable}
cd /tmp
for S in $STAGES; do
case $S in
bench)
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_steps20.json 2> $O/bench_driver_form.err
  python3 $R/bench.py > $O/bench_default_all_configs.json 2> $O/bench_default.err
  echo "bench done";;
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/driver_stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_profiled_run.json 2> $O/driver_stats.err || true
  echo "stats done";;
pmc)
  H="--other-configs off --steps 64 --warmup 16 --compare-steps 0 --host-path-steps 0 --live-loop-steps 0 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/n16384_fetch -- python3 $R/bench.py $H > /dev/null 2> $O/n16384_fetch.err || true
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/n16384_write -- python3 $R/bench.py $H > /dev/null 2> $O/n16384_write.err || true
  echo "headline pmc done";;
stable)
  W="--workload n16384-ellstable --steps 48 --warmup 8 --host-path-steps 0 --live-loop-steps 0 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stable_stats -- python3 $R/bench.py $W > $O/bench_ellstable_profiled_run.json 2> $O/stable_stats.err || true
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/stable_fetch -- python3 $R/bench.py $W > /dev/null 2> $O/stable_fetch.err || true
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/stable_write -- python3 $R/bench.py $W > /dev/null 2> $O/stable_write.err || true
  echo "ellstable done";;
esac
done
cd $R
[ -d $O/n16384_fetch ] && [ -d $O/driver_stats ] && python3 tools/pmc_summary.py $O/driver_stats $O/n16384_fetch $O/n16384_write $O/summary_n16384 bench_driver_form n16384-parallel || true
[ -d $O/stable_fetch ] && python3 tools/pmc_summary.py $O/stable_stats $O/stable_fetch $O/stable_write $O/summary_ellstable bench_ellstable n16384-ellstable || true
[ -d $O/driver_stats ] && python3 tools/trace_by_grid.py $O/driver_stats $O/bench_driver_form_kernel_stats_by_grid.csv || true
for f in $O/bench_driver_form_steps20.json $O/bench_default_all_configs.json; do [ -f $f ] && python3 tools/show_bench.py $f | cut -c1-260; done
true
