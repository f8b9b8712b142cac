#!/bin/bash
# The Cin-split knobs on the two small-batch steps, one process per setting, two rounds: tools/split_knob_sweep.sh
for r in 1 2; do
  for kn in "" "DM3D_CONV_SPLIT_MAXPARTS=8" "DM3D_CONV_SPLIT_MAXPARTS=4" "DM3D_CONV_SPLIT_TARGET=256" "DM3D_CONV_SPLIT_TARGET=1024" \
            "DM3D_CONV_SPLIT_MINCHUNKS=1" "DM3D_CONV_SPLIT_MINCHUNKS=4" "DM3D_CONV_SPLIT_TARGET=256 DM3D_CONV_SPLIT_MAXPARTS=8" "DM3D_CONV_SPLIT_WGS=128"; do
    for cfg in "--batch 4 --channels 4" "--batch 1 --channels 8"; do
      env $kn python tools/bench_kinds.py "[${kn:-defaults}] [$cfg]" $cfg --steps 50 --warmup 5 || exit 1
    done
  done
done
