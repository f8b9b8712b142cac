#!/bin/bash
# Compiles one csrc/*.hip to ISA and lists, per kernel, the basic blocks that hold MFMAs together with scratch (spill) traffic or vmcnt(0).
#   tools/isa_check.sh dm3d_conv_h3w.hip [extra flags]
f=$1; shift
cd "$(dirname "$0")/../3d-condtional-stable-diffusion_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -ffp-contract=off -S --cuda-device-only "$@" $f -o /tmp/${f%.hip}.s 2>/dev/null
python3 ../../tools/isa_blocks.py /tmp/${f%.hip}.s | awk '/^_Z/ || (/mfma +[1-9]/ && (/scratch_load +[1-9]/ || /scratch_store +[1-9]/ || /vmcnt\(0\) +[1-9]/))'
