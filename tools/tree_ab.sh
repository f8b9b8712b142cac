#!/bin/bash
# Interleaved A/B of two whole source trees on one box: tools/tree_ab.sh <other tree> "<bench.py args>" [rounds]   (e.g. ab/r04 = `git archive` of an
# older commit with its built library inside; gpurun ships it with the snapshot).  Prints ms_per_step and the per-kind table of each run.
other=$1; args=$2; rounds=${3:-2}
here=$PWD
for r in $(seq 1 $rounds); do for t in $other $here; do
  (cd $t && python3 bench.py $args --no-cpu-baseline --no-fp32-mode --no-full-chain 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['per_kernel_kind']
print('$t'.split('/')[-1].ljust(8), round(d['ms_per_step'],3), ' '.join(f\"{x}:{k[x]['launches_per_step']}x={k[x]['ms_per_step']:.3f}\" for x in sorted(k)))")
done; done
