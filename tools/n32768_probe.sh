set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
W="--workload n32768-deep --steps 64 --warmup 16 --compare-steps 0 --host-path-steps 0 --no-cpu-baseline"
python3 $R/bench.py $W --opt QUEUE_DEPTH=0 > $O/b32k_q0.json 2>/dev/null
python3 $R/tools/show_bench.py $O/b32k_q0.json | cut -c1-300
python3 $R/bench.py $W > $O/b32k_q48.json 2>/dev/null
python3 $R/tools/show_bench.py $O/b32k_q48.json | cut -c1-300
rocprofv3 --kernel-trace --output-format csv -d $O/b32k_trace -- python3 $R/bench.py $W > /dev/null 2> $O/b32k_trace.err
python3 $R/tools/trace_by_grid.py $O/b32k_trace | head -30
