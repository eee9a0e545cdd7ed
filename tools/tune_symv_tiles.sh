#!/bin/bash
# Tuning builds of k_symv tile shapes (run on the GPU box): rebuilds libellhip.so with -D overrides and times
# the default bench.  Restores nothing: the box is scratch.
set -u
for cfg in ${SYMV_TILE_CFGS:-"2048 64 2" "4096 32 2" "4096 32 1" "8192 16 1" "2048 32 2" "2048 32 4"}; do
  set -- $cfg
  ELLHIP_EXTRA_HIPCC_FLAGS="-DELLHIP_SYMV_SEG=$1 -DELLHIP_SYMV_H=$2" python -c "
import importlib, sys
sys.path.insert(0, '.')
b = importlib.import_module('ellalgo-rs_amd.build'); b.build(force=True)" || exit 1
  ELLHIP_SYMV_RW=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --host-path-steps 0 --compare-steps 0 --steps 160 > gpurun_out/symv_tile.json 2> gpurun_out/symv_tile.err || { tail -3 gpurun_out/symv_tile.err; continue; }
  echo "SEG=$1 H=$2 RW=$3: $(python tools/show_bench.py gpurun_out/symv_tile.json | head -1)"
done
