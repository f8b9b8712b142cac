#!/bin/bash
# A/B variant libraries on the GEMM micro-benchmark.  usage: tools/abg.sh v0 va vb ...
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v (round $round)"
    DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$v.so python tools/gemm_bench.py 2>&1 | grep -E " us "
  done
done
