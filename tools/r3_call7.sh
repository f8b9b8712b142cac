#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/r3_ops_skip.log 2>&1 || { tail -30 $out/r3_ops_skip.log; exit 1; }
tail -2 $out/r3_ops_skip.log
for B in 4 1 32; do
  timeout -k 10 300 python3 tools/layer_profile.py h3 $B > $out/r3_layers_B${B}_skip.log 2>&1 || tail -20 $out/r3_layers_B${B}_skip.log
  tail -1 $out/r3_layers_B${B}_skip.log
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_unet.py -x -q -m gpu > $out/r3_unet_skip.log 2>&1 || { tail -30 $out/r3_unet_skip.log; exit 1; }
tail -2 $out/r3_unet_skip.log
