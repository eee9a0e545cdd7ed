#!/bin/bash
# One rocprofv3 PMC pass with SQ counters (no other tracing but the kernel trace) over the headline queue run: where do the waves of
# k_symm_mfma / k_apply_mfma spend their cycles?  WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA RAW / pipe)
# + ACTIVE_INST_ANY ~ WAVE_CYCLES.  Usage (GPU box): tools/pmc_sq_pass.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
H="--other-configs off --steps 96 --warmup 48 --compare-steps 0 --host-path-steps 0 --live-loop-steps 0 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py $H > /dev/null 2> $O/sq.err || true
cd $R
python3 - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/sq/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ellhip::k_" not in k: continue
        short = k.split("ellhip::")[1].split("(")[0]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$O/sq_summary.txt", "w")
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"]
print(f"{'kernel':34s} launches " + " ".join(f"{n[3:]:>22s}" for n in names), file=out)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    row = [sum(v[n]) / max(1, len(v[n])) for n in names]
    wc = row[0] or 1.0
    print(f"{k[:34]:34s} {len(v[names[0]]):8d} " + " ".join(f"{x:14.4g} ({x / wc:5.2f})" for x in row), file=out)
out.close()
print(open("$O/sq_summary.txt").read())
PY
