#!/bin/bash
# A/B of a variant library against the product library on one box, interleaved:  tools/ab_lib.sh <variant name> [rounds]
# (variants are built by tools/mk_variant.sh <name> <flags>)
v=$1; rounds=${2:-3}
for round in $(seq 1 $rounds); do
  for which in product $v; do
    if [ $which = product ]; then unset DM3D_LIB; else export DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/$which.so; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-full-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['per_kernel_kind']
print('$which round $round', round(d['ms_per_step'],3), 'ms/step;', ' '.join(f'{n} {v[\"ms_per_step\"]}' for n,v in k.items() if n.startswith(('conv','gemm','attn'))))"
  done
done
