#!/bin/bash
# rocprofv3 kernel stats for BASELINE config 5 (n = 16384 EllStable), beside the bench line of the same command.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_stats -- python3 $R/bench.py --workload n16384-ellstable --steps 100 --warmup 10 --no-cpu-baseline --host-path-steps 0 > $O/st_stats.json 2> $O/st_stats.err
cd $R
cp $O/st_stats/*/*_kernel_stats.csv $O/bench_n16384-ellstable_kernel_stats.csv
head -9 $O/bench_n16384-ellstable_kernel_stats.csv | cut -c1-150
python tools/show_bench.py $O/st_stats.json | head -2
