#!/bin/bash
# A/B of an environment switch on one box, interleaved:  tools/ab_env.sh VAR v0 v1 [rounds]   (bench.py steps, no CPU baseline)
var=$1; a=$2; b=$3; rounds=${4:-3}
for round in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-full-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['per_kernel_kind']
print('$var=$v round $round', round(d['ms_per_step'],3), 'ms/step;', ' '.join(f'{n} {v[\"ms_per_step\"]}' for n,v in k.items() if n.startswith(('conv','gemm'))))"
  done
done
