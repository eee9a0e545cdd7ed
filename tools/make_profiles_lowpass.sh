#!/bin/bash
# Evidence for the LowpassOracle row on the GPU box: bench JSONs, rocprofv3 kernel stats and the PMC passes.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --workload lowpass-n4096 --steps 2000 --warmup 200 > $O/bench_lowpass_n4096.json 2> $O/bench_lowpass_n4096.err
python bench.py --workload lowpass-n1024 --steps 2000 --warmup 200 > $O/bench_lowpass_n1024.json 2>/dev/null
python bench.py --workload lowpass-n4096 --no-cpu-baseline > $O/bench_lowpass_n4096_first220.json 2>/dev/null
export TMPDIR=/tmp
ARGS="bench.py --workload lowpass-n4096 --steps 2000 --warmup 200 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lp_stats -- python3 $R/$ARGS > $O/lp_stats.json 2> $O/lp_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/lp_fetch -- python3 $R/$ARGS > /dev/null 2> $O/lp_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/lp_write -- python3 $R/$ARGS > /dev/null 2> $O/lp_write.err
cd $R
python tools/pmc_summary.py $O/lp_stats $O/lp_fetch $O/lp_write $O/lp_summary bench_lowpass_n4096 lowpass-n4096
python tools/show_bench.py $O/bench_lowpass_n4096.json $O/bench_lowpass_n1024.json $O/bench_lowpass_n4096_first220.json $O/lp_stats.json
