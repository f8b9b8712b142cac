#!/bin/bash
# Regenerates the evidence under profiles/ for the H3 path on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r01_h3
# 1. bench.py default run (with cpu_baseline)            -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same bench  -> gpurun_out/<tag>_kernel_stats.csv
# 3. two --pmc passes (FETCH_SIZE, WRITE_SIZE), no graph -> gpurun_out/<tag>_pmc_hbm.csv  (profiles/summarize_pmc.py)
# Copy the three files into profiles/ afterwards (gpurun_out/ is scratch).
set -e
tag=${1:-r02_h3}
prec=${2:-h3}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
python3 bench.py --precision $prec --steps 30 --warmup 5 2> $out/${tag}_bench.log | tail -1 > $out/${tag}_bench.json
echo "bench done: $(cut -c1-160 $out/${tag}_bench.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -o run -- python3 bench.py --precision $prec --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-full-chain > $out/prof_stats.log 2>&1
cp $(find $out/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
echo "stats done: $(sed -n 2p $out/${tag}_kernel_stats.csv | cut -c1-200)"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/prof_$c -o run -- python3 bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-full-chain --no-graph > $out/prof_$c.log 2>&1
  mkdir -p $out/pmc_$c && cp $(find $out/prof_$c -name "*counter_collection.csv" | head -1) $out/pmc_$c/pmc_counter_collection.csv
done
python3 profiles/summarize_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/${tag}_pmc_hbm.csv \
  "$tag: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph (B=32, 32^3x8ch); KB per launch" \
  "batch=32 size=32 channels=8 norm=batch precision=$prec csrc=$(python3 bench.py --print-csrc-digest)" | head -8
# 4. SQ counters of the same command (two passes, 8 SQ slots each) -> <tag>_sq.csv (profiles/summarize_sq.py)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/prof_sqA -o run -- python3 bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-full-chain --no-graph > $out/prof_sqA.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/prof_sqB -o run -- python3 bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-full-chain --no-graph > $out/prof_sqB.log 2>&1
python3 profiles/summarize_sq.py $out/${tag}_sq.csv \
  "$tag: rocprofv3 --pmc <8 SQ counters> GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph (B=32, 32^3x8ch; two passes); averages per launch" \
  "batch=32 size=32 channels=8 norm=batch precision=$prec csrc=$(python3 bench.py --print-csrc-digest)" $out/prof_sqA $out/prof_sqB | head -6 | cut -c1-400
# 5. in-kernel clock of the conv kernel (diagnostic library variants/cck.so: python3 tools/mk_stamp_variants.py) -> <tag>_clock.csv
if [ -f 3d-condtional-stable-diffusion_amd/csrc/variants/cck.so ]; then
  # <tag>_clock.csv: the Winograd-x form (the dominant kernel at B = 32 since round 3); <tag>_clockv3.csv: the direct free-running kernel
  lib=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/cck.so
  CLOCK_WINO=1 CLOCK_OUT=$out/${tag}_clock.csv DM3D_LIB=$lib python3 tools/kernel_clock.py conv > $out/${tag}_clock.log 2>&1 || tail -5 $out/${tag}_clock.log
  CLOCK_WINO=1 CLOCK_ZEROS=1 CLOCK_OUT=$out/${tag}_clock_zeros.txt DM3D_LIB=$lib python3 tools/kernel_clock.py conv > $out/${tag}_clock_zeros.log 2>&1 || true
  DM3D_CONV_WINO=0 CLOCK_OUT=$out/${tag}_clockv3.csv DM3D_LIB=$lib python3 tools/kernel_clock.py conv > $out/${tag}_clockv3.log 2>&1 || tail -5 $out/${tag}_clockv3.log
  DM3D_CONV_WINO=0 CLOCK_ZEROS=1 CLOCK_OUT=$out/${tag}_clockv3_zeros.txt DM3D_LIB=$lib python3 tools/kernel_clock.py conv > $out/${tag}_clockv3_zeros.log 2>&1 || true
  cat $out/${tag}_clock.csv $out/${tag}_clockv3.csv
fi
rm -rf $out/prof_stats $out/prof_FETCH_SIZE $out/prof_WRITE_SIZE $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/prof_sqA $out/prof_sqB
