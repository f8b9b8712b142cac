#!/bin/bash
# bench.py under two settings of one environment variable, alternating, on one box:  tools/bench_ab.sh VAR v0 v1 [kinds regex]
var=$1; a=$2; b=$3; kinds=${4:-"gemm_h3|mlp_fused|attn_fused|layernorm"}
for r in 1 2; do for v in $a $b; do
  env $var=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-mode --no-full-chain 2>gpurun_out/bench_ab.err | python -c "
import json,sys,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['per_kernel_kind']
print('$var=$v', round(d['ms_per_step'],3), {x:(k[x]['launches_per_step'],k[x]['ms_per_step'],k[x]['tflops']) for x in k if re.search(r'$kinds', x)})" || tail -3 gpurun_out/bench_ab.err
done; done
