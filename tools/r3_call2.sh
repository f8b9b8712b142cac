#!/bin/bash
# round 3, GPU call 2: first run of the free-running conv kernel (v3): op tests, A/B vs v2, clocks
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $out/r3_v3_ops.log 2>&1 || { tail -30 $out/r3_v3_ops.log; exit 1; }
tail -3 $out/r3_v3_ops.log
timeout -k 10 200 python3 tools/env_ab.py DM3D_CONV_V3 0 1 > $out/r3_v3_ab.log 2>&1 || { tail -20 $out/r3_v3_ab.log; exit 1; }
cat $out/r3_v3_ab.log
DM3D_CONV_V3_TD=8 timeout -k 10 200 python3 tools/env_ab.py DM3D_CONV_V3 0 1 > $out/r3_v3_ab_td8.log 2>&1 || tail -20 $out/r3_v3_ab_td8.log
cat $out/r3_v3_ab_td8.log
timeout -k 10 300 python3 tools/layer_profile.py h3 32 > $out/r3_v3_layers.log 2>&1 || tail -20 $out/r3_v3_layers.log
tail -1 $out/r3_v3_layers.log
DM3D_CONV_V3=0 timeout -k 10 300 python3 tools/layer_profile.py h3 32 > $out/r3_v2_layers.log 2>&1 || tail -20 $out/r3_v2_layers.log
tail -1 $out/r3_v2_layers.log
