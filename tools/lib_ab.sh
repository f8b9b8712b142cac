#!/bin/bash
# Interleaved A/B of library builds on one box:  tools/lib_ab.sh "<python script + args>" libA.so libB.so ...   ("-" = the product library)
# Each build runs ROUNDS (default 2) times, alternating, so that thermal drift shows up as a difference between rounds, not between builds.
cmd=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then env -u DM3D_LIB python $cmd product || exit 1
    else DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$lib python $cmd ${lib%.so} || exit 1; fi
  done
done
