#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_vqvae.py -x -q -m gpu > $out/r3_nct_ops.log 2>&1 || { tail -40 $out/r3_nct_ops.log; exit 1; }
tail -2 $out/r3_nct_ops.log
for B in 32 4; do
  timeout -k 10 300 python3 tools/layer_profile.py h3 $B > $out/r3_layers_B${B}_nct.log 2>&1 || tail -20 $out/r3_layers_B${B}_nct.log
  grep "n32\|total" $out/r3_layers_B${B}_nct.log
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_unet.py tests/test_gpu_round3.py -x -q -m gpu > $out/r3_nct_unet.log 2>&1 || { tail -40 $out/r3_nct_unet.log; exit 1; }
tail -2 $out/r3_nct_unet.log
