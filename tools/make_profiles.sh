#!/bin/bash
# Regenerate the round's evidence on the GPU box: bench JSONs, rocprofv3 kernel stats and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, one counter per pass) for the headline workload.  Outputs under gpurun_out/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ -z "$PROFILES_ONLY" ]; then
python bench.py > $O/bench_n16384.json 2> $O/bench_n16384.err
for w in n8192-deep n32768-deep n4096-deep; do
  python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null
done
fi
export TMPDIR=/tmp
ARGS="bench.py --steps 64 --warmup 16 --compare-steps 0 --host-path-steps 0 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/$ARGS > $O/prof_stats.json 2> $O/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 $R/$ARGS > /dev/null 2> $O/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 $R/$ARGS > /dev/null 2> $O/prof_write.err
cd $R
python tools/pmc_summary.py $O/prof_stats $O/prof_fetch $O/prof_write $O/summary bench_n16384 n16384-parallel
[ -n "$PROFILES_ONLY" ] || python tools/show_bench.py $O/bench_n16384.json $O/bench_n8192-deep.json $O/bench_n32768-deep.json $O/bench_n4096-deep.json
