#!/bin/bash
# rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) for ONE bench workload.
# Usage (on the GPU box): tools/profile_workload.sh <workload> <tag> [extra bench args]; outputs under gpurun_out/<tag>_*
set -e
W=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
ARGS="bench.py --workload $W --steps 64 --warmup 16 --compare-steps 0 --host-path-steps 0 --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/$ARGS > $O/${TAG}_profiled_run.json 2> $O/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -- python3 $R/$ARGS > /dev/null 2> $O/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -- python3 $R/$ARGS > /dev/null 2> $O/${TAG}_write.err
cd $R
python tools/pmc_summary.py $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_summary bench_$TAG $W
