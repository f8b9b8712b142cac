#!/bin/bash
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
AB_SMALL=1 python3 tools/env_ab.py DM3D_CONV_STAGGER 0 3 6 12 24 > $out/r3_stagger_b32.log 2>&1 || tail -5 $out/r3_stagger_b32.log
cat $out/r3_stagger_b32.log
AB_BATCH=4 python3 tools/env_ab.py DM3D_CONV_STAGGER 0 6 12 > $out/r3_stagger_b4.log 2>&1 || tail -5 $out/r3_stagger_b4.log
cat $out/r3_stagger_b4.log
AB_BATCH=4 AB_SMALL=1 python3 tools/env_ab.py DM3D_CONV_STAGGER 0 6 12 > $out/r3_stagger_b4s.log 2>&1 || tail -5 $out/r3_stagger_b4s.log
cat $out/r3_stagger_b4s.log
