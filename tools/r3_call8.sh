#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $out/r3_ops_skip2.log 2>&1 || { tail -30 $out/r3_ops_skip2.log; exit 1; }
tail -2 $out/r3_ops_skip2.log
for B in 4 1; do
  timeout -k 10 300 python3 tools/layer_profile.py h3 $B > $out/r3_layers_B${B}_skip2.log 2>&1 || tail -20 $out/r3_layers_B${B}_skip2.log
  tail -1 $out/r3_layers_B${B}_skip2.log
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_unet.py -x -q -m gpu > $out/r3_unet_skip2.log 2>&1 || { tail -30 $out/r3_unet_skip2.log; exit 1; }
tail -2 $out/r3_unet_skip2.log
python3 bench.py --batch 4 --channels 4 --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-h3f8-mode 2> /dev/null | tail -1 > $out/r3_config2_bench.json
python3 -c "
import json; d=json.load(open('$out/r3_config2_bench.json')); print('config2', d['ms_per_step'], d['value'])"
python3 bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-h3f8-mode --no-full-chain 2> /dev/null | tail -1 > $out/r3_b1_bench.json
python3 -c "
import json; d=json.load(open('$out/r3_b1_bench.json')); print('B=1', d['ms_per_step'], d['value'])"
