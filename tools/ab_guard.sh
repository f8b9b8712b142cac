#!/bin/bash
# A/B of the H3 range guard on one box: bench.py steps with the product library and with the -DDM3D_NO_RANGE_GUARD variant
# (tools/mk_variant.sh noguard -DDM3D_NO_RANGE_GUARD), interleaved.
for round in 1 2 3; do
  for v in product noguard; do
    if [ $v = product ]; then unset DM3D_LIB; else export DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$v.so; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-full-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['per_kernel_kind']
print('$v round $round', round(d['ms_per_step'],3), 'ms/step; conv_k3s1', k['conv_k3s1']['ms_per_step'], 'h2in', k['conv_k3s1_h2in']['ms_per_step'], 'gemm_h3', k['gemm_h3']['ms_per_step'])"
  done
done
