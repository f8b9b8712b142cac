#!/bin/bash
# The end-of-round evidence run (one gpurun call): GPU tests, tools/profile_round.sh, then the bench lines that attach the fresh summaries.
#   gpurun --timeout 1200 -- 'bash tools/final_round.sh r05'
set -e
tag=${1:-r05}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_tests.log 2>&1 || { tail -40 $out/${tag}_gpu_tests.log; exit 1; }
tail -2 $out/${tag}_gpu_tests.log
bash tools/profile_round.sh ${tag}_h3 h3
for f in pmc_hbm sq clock clockv3; do [ -f $out/${tag}_h3_$f.csv ] && cp $out/${tag}_h3_$f.csv profiles/; done      # so that the bench below attaches them
[ -f $out/${tag}_h3_clock.csv ] || echo "WARNING: no in-kernel clock summary (csrc/variants/cck.so missing: python tools/mk_stamp_variants.py first)"
python3 bench.py --steps 30 --warmup 5 2> $out/${tag}_h3_bench.log | tail -1 > $out/${tag}_h3_bench.json
python3 bench.py --batch 4 --channels 4 --steps 50 --warmup 5 --no-fp32-mode 2> /dev/null | tail -1 > $out/${tag}_h3_config2_bench.json
python3 bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-full-chain 2> /dev/null | tail -1 > $out/${tag}_h3_b1_bench.json
python3 bench.py --norm group --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-mode --no-full-chain 2> /dev/null | tail -1 > $out/${tag}_h3_groupnorm_bench.json
python3 tools/e2e_config5.py > $out/${tag}_e2e_config5.log 2>&1 || tail -5 $out/${tag}_e2e_config5.log
python3 tools/chain_repeatability.py > $out/${tag}_chain_repeatability.log 2>&1 || tail -5 $out/${tag}_chain_repeatability.log
python3 tools/small_batch_floor.py > $out/${tag}_small_batch_floor.log 2>&1 || tail -5 $out/${tag}_small_batch_floor.log
python3 -c "
import json
for f in ('h3_bench','h3_config2_bench','h3_b1_bench','h3_groupnorm_bench'):
    d=json.load(open('$out/${tag}_'+f+'.json')); r=d.get('roofline') or {}
    print(f, round(d['ms_per_step'],3), round(d['value'],4), r.get('achieved'), r.get('traffic'), r.get('mfma_busy_frac'), r.get('clock_ghz'))"
tail -3 $out/${tag}_e2e_config5.log; tail -3 $out/${tag}_chain_repeatability.log
