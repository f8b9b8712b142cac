#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $out/r3_tests_round3.log 2>&1 || { tail -60 $out/r3_tests_round3.log; }
tail -5 $out/r3_tests_round3.log
for B in 4 1; do
  timeout -k 10 300 python3 tools/layer_profile.py h3 $B > $out/r3_layers_B$B.log 2>&1 || tail -20 $out/r3_layers_B$B.log
  tail -1 $out/r3_layers_B$B.log
done
python3 bench.py --batch 4 --channels 4 --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-h3f8-mode 2> /dev/null | tail -1 > $out/r3_config2_bench.json
python3 -c "
import json; d=json.load(open('$out/r3_config2_bench.json')); print('config2', d['ms_per_step'], d['value'])"
