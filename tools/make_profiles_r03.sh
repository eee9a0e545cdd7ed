#!/bin/bash
# Round-3 evidence on the GPU box: the default bench line (all BASELINE configurations in one line), the rocprofv3
# kernel-trace summary of THAT command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass, no other
# tracing) for the headline workload and for the resident n = 4096 run.  Outputs under gpurun_out/r3prof/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3prof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default_stats -- python3 $R/bench.py > $O/bench_default_profiled_run.json 2> $O/default_stats.err
echo "stats done"
H="--other-configs off --steps 64 --warmup 16 --compare-steps 0 --host-path-steps 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/n16384_fetch -- python3 $R/bench.py $H > /dev/null 2> $O/n16384_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/n16384_write -- python3 $R/bench.py $H > /dev/null 2> $O/n16384_write.err
echo "headline pmc done"
W="--workload n4096-deep --steps 200 --warmup 20 --compare-steps 0 --host-path-steps 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n4096_stats -- python3 $R/bench.py $W > $O/bench_n4096_profiled_run.json 2> $O/n4096_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/n4096_fetch -- python3 $R/bench.py $W > /dev/null 2> $O/n4096_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/n4096_write -- python3 $R/bench.py $W > /dev/null 2> $O/n4096_write.err
echo "n4096 done"
cd $R
python tools/pmc_summary.py $O/default_stats $O/n16384_fetch $O/n16384_write $O/summary_n16384 bench_n16384 n16384-parallel || true
python tools/pmc_summary.py $O/n4096_stats $O/n4096_fetch $O/n4096_write $O/summary_n4096 bench_n4096 n4096-deep || true
python tools/trace_by_grid.py $O/default_stats $O/bench_default_kernel_stats_by_grid.csv || true
python tools/show_bench.py $O/bench_default.json | cut -c1-300
