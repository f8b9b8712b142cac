#!/bin/bash
# Interleaved A/B of library builds on the two small-batch steps (config 2 = 32^3 x 4ch B = 4, and B = 1 of config 3's shape):
#   tools/ab_small_batch.sh libA.so libB.so ...      ("-" = the product library; others are looked up in csrc/variants/)
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    for cfg in "--batch 4 --channels 4" "--batch 1 --channels 8"; do
      if [ "$lib" = "-" ]; then env -u DM3D_LIB python tools/bench_kinds.py "product [$cfg]" $cfg --steps 50 --warmup 5 || exit 1
      else DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$lib python tools/bench_kinds.py "${lib%.so} [$cfg]" $cfg --steps 50 --warmup 5 || exit 1; fi
    done
  done
done
