#!/bin/bash
# A/B the variant libraries under 3d-condtional-stable-diffusion_amd/csrc/variants on one box, interleaved rounds.
# usage: tools/ab.sh "<grep pattern>" v0 va vb ...
pat="$1"; shift
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v (round $round)"
    DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$v.so python tools/conv_bench.py h3 32 2>&1 | grep -E "$pat"
  done
done
