#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the C++ live loop (host/bench/live_loop) and its median per-iteration timeline.
# Usage (on the GPU box): tools/live_loop_trace.sh <tag> [n] [warm] [steps]; outputs under gpurun_out/<tag>*
set -e
TAG=$1; N=${2:-16384}; WARM=${3:-48}; STEPS=${4:-200}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
CUTS=/dev/shm/ellhip_live_cuts_$$.bin
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from ellalgo_rs_amd import synth
k, g, b0, b1 = synth.parallel_cuts($N, $WARM + $STEPS + 1)
synth.write_cuts_bin("$CUTS", k, g, b0, b1)
PY
cd /tmp
$R/ellalgo-rs_amd/host/bench/live_loop $CUTS $WARM $STEPS > $O/${TAG}_untraced.json
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/${TAG}_trace -- $R/ellalgo-rs_amd/host/bench/live_loop $CUTS $WARM $STEPS > $O/${TAG}_traced.json 2> $O/${TAG}_trace.err
rm -f $CUTS
cd $R
python3 tools/trace_timeline.py $O/${TAG}_trace > $O/${TAG}_timeline.txt
cat $O/${TAG}_untraced.json $O/${TAG}_timeline.txt
