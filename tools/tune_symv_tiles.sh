#!/bin/bash
# Tuning builds of k_symv tile shapes (run on the GPU box): rebuilds libellhip.so with -D overrides and times
# the default bench.  Restores nothing: the box is scratch.
set -u
for cfg in "2048 64 2" "1024 64 2" "1024 64 4" "1024 128 4" "2048 128 2" "512 128 8" "1024 128 2"; do
  set -- $cfg
  ELLHIP_EXTRA_HIPCC_FLAGS="-DELLHIP_SYMV_SEG=$1 -DELLHIP_SYMV_H=$2" python -c "
import importlib, sys
sys.path.insert(0, '.')
b = importlib.import_module('ellalgo-rs_amd.build'); b.build(force=True)" || exit 1
  ELLHIP_SYMV_RW=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --host-path-steps 0 --compare-steps 0 --steps 160 > gpurun_out/symv_tile.json 2> gpurun_out/symv_tile.err || { tail -3 gpurun_out/symv_tile.err; continue; }
  echo "SEG=$1 H=$2 RW=$3: $(python tools/show_bench.py gpurun_out/symv_tile.json | head -1)"
done
