#!/bin/bash
# Tuning builds of the scalar stage's slice length (run on the GPU box; the box is scratch).
for sl in 1024 512 256; do
  ELLHIP_EXTRA_HIPCC_FLAGS="-DELLHIP_SCALAR_SLICE=$sl" python -c "
import importlib, sys
sys.path.insert(0, '.')
b = importlib.import_module('ellalgo-rs_amd.build'); b.build(force=True)" || exit 1
  for w in n4096-deep n8192-deep n16384-parallel; do
    timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --host-path-steps 0 --compare-steps 0 > gpurun_out/slice.json 2> gpurun_out/slice.err || { tail -3 gpurun_out/slice.err; continue; }
    echo "slice=$sl: $(python tools/show_bench.py gpurun_out/slice.json | head -1)"
  done
done
