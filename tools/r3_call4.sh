#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/r3_gpu_tests_v3.log 2>&1 || { tail -40 $out/r3_gpu_tests_v3.log; exit 1; }
tail -3 $out/r3_gpu_tests_v3.log
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> $out/r3_bench_v3.err | tail -1 > $out/r3_bench_v3.json
python3 -c "
import json; d=json.load(open('$out/r3_bench_v3.json')); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], {k:v['ms_per_step'] for k,v in d['per_kernel_kind'].items()}, d['fp32_mode']['ms_per_step'], d['h3f8_mode']['ms_per_step'])"
