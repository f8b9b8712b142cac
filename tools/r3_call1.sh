#!/bin/bash
# round 3, GPU call 1: in-kernel clocks, SQ counters of the dominant conv, weight-ring depth A/B
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
V=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants
python3 tools/env_ab.py DM3D_CONV_RING 3 4 5 > $out/r3_ring_ab.log 2>&1 || { tail -20 $out/r3_ring_ab.log; exit 1; }
cat $out/r3_ring_ab.log
./tools/micro/mfma_shapes > $out/r3_mfma_clock.log 2>&1; tail -4 $out/r3_mfma_clock.log
for ring in 3 5; do
  DM3D_CONV_RING=$ring DM3D_LIB=$V/cck.so python3 tools/kernel_clock.py conv > $out/r3_clock_conv_ring$ring.log 2>&1 || { tail -20 $out/r3_clock_conv_ring$ring.log; exit 1; }
  cat $out/r3_clock_conv_ring$ring.log
done
DM3D_LIB=$V/gst.so python3 tools/kernel_clock.py gemm > $out/r3_clock_gemm.log 2>&1 || tail -20 $out/r3_clock_gemm.log
cat $out/r3_clock_gemm.log
DM3D_LIB=$V/cst.so python3 tools/conv_stamps.py > $out/r3_stamps_ring3.log 2>&1 || tail -5 $out/r3_stamps_ring3.log
DM3D_CONV_RING=5 DM3D_LIB=$V/cst.so python3 tools/conv_stamps.py > $out/r3_stamps_ring5.log 2>&1 || tail -5 $out/r3_stamps_ring5.log
for ring in 3 5; do
  export DM3D_CONV_RING=$ring
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r3_sqA_ring$ring -o p -- python3 tools/conv_pmc.py h3 pro192 > $out/r3_sqA_ring$ring.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r3_sqB_ring$ring -o p -- python3 tools/conv_pmc.py h3 pro192 > $out/r3_sqB_ring$ring.log 2>&1
  python3 profiles/summarize_sq.py $out/r3_sq_conv_ring$ring.csv "rocprofv3 --pmc <SQ counters> GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/conv_pmc.py h3 pro192 (32^3 192->64, norm+SiLU prologue; two passes; B=32; DM3D_CONV_RING=$ring)" "tools/conv_pmc.py h3 pro192 csrc=$(python3 bench.py --print-csrc-digest)" $out/r3_sqA_ring$ring $out/r3_sqB_ring$ring
done
unset DM3D_CONV_RING
DM3D_CONV_RING=5 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $out/r3_ring5_tests.log 2>&1 || true
tail -3 $out/r3_ring5_tests.log
