#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 300 python3 tools/layer_profile.py h3 32 > $out/r3_layers_B32_nct8.log 2>&1 || tail -20 $out/r3_layers_B32_nct8.log
grep "n32\|total" $out/r3_layers_B32_nct8.log
DM3D_CONV_V3_TD=4 timeout -k 10 300 python3 tools/layer_profile.py h3 32 > $out/r3_layers_B32_nct4.log 2>&1 || tail -20 $out/r3_layers_B32_nct4.log
grep "n32\|total" $out/r3_layers_B32_nct4.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k conv > $out/r3_nct8_ops.log 2>&1 || { tail -40 $out/r3_nct8_ops.log; exit 1; }
tail -2 $out/r3_nct8_ops.log
