#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or attention or dense" > $out/r3_gemm_tests_auto.log 2>&1 || { tail -30 $out/r3_gemm_tests_auto.log; exit 1; }
tail -2 $out/r3_gemm_tests_auto.log
DM3D_GEMM_MR=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or attention or dense" > $out/r3_gemm_tests_mr1.log 2>&1 || { tail -30 $out/r3_gemm_tests_mr1.log; exit 1; }
tail -2 $out/r3_gemm_tests_mr1.log
for B in 4 1; do
  timeout -k 10 300 python3 tools/layer_profile.py h3 $B > $out/r3_layers_B${B}_gemm.log 2>&1 || tail -20 $out/r3_layers_B${B}_gemm.log
  tail -1 $out/r3_layers_B${B}_gemm.log
done
for mr in 2 1; do
  DM3D_GEMM_MR=$mr timeout -k 10 300 python3 tools/layer_profile.py h3 32 > $out/r3_layers_B32_mr$mr.log 2>&1 || tail -20 $out/r3_layers_B32_mr$mr.log
  grep -c . $out/r3_layers_B32_mr$mr.log; grep "gemm_h3\|total" $out/r3_layers_B32_mr$mr.log | awk '{s+=$1} END {print "gemm+total sum", s}'
done
./tools/micro/mfma_dtypes > $out/r3_mfma_dtypes.log 2>&1; cat $out/r3_mfma_dtypes.log
