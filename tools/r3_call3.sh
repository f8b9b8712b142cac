#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
V=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants
DM3D_LIB=$V/cck.so python3 tools/kernel_clock.py conv > $out/r3_clock_v3_td4.log 2>&1 || { tail -20 $out/r3_clock_v3_td4.log; exit 1; }
cat $out/r3_clock_v3_td4.log
DM3D_CONV_V3_TD=8 DM3D_LIB=$V/cck.so python3 tools/kernel_clock.py conv > $out/r3_clock_v3_td8.log 2>&1 || { tail -20 $out/r3_clock_v3_td8.log; exit 1; }
cat $out/r3_clock_v3_td8.log
CLOCK_ZEROS=1 DM3D_LIB=$V/cck.so python3 tools/kernel_clock.py conv > $out/r3_clock_v3_td4_zeros.log 2>&1 || { tail -20 $out/r3_clock_v3_td4_zeros.log; exit 1; }
cat $out/r3_clock_v3_td4_zeros.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r3_sqA_v3 -o p -- python3 tools/conv_pmc.py h3 pro192 > $out/r3_sqA_v3.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r3_sqB_v3 -o p -- python3 tools/conv_pmc.py h3 pro192 > $out/r3_sqB_v3.log 2>&1
python3 profiles/summarize_sq.py $out/r3_sq_conv_v3.csv "rocprofv3 --pmc <SQ counters> GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/conv_pmc.py h3 pro192 (32^3 192->64, norm+SiLU prologue; two passes; B=32)" "tools/conv_pmc.py h3 pro192 csrc=$(python3 bench.py --print-csrc-digest)" $out/r3_sqA_v3 $out/r3_sqB_v3 | head -4 | cut -c1-600
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/r3_hbm_$c -o p -- python3 tools/conv_pmc.py h3 pro192 > $out/r3_hbm_$c.log 2>&1
  grep -h "h3v3" $out/r3_hbm_$c/*/*counter_collection.csv $out/r3_hbm_$c/*counter_collection.csv 2>/dev/null | awk -F, '{print $(NF-3), $(NF-2)}' | tail -3
done
