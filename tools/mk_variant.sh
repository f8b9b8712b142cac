#!/bin/bash
# Builds an A/B variant of the library with extra compiler flags:  tools/mk_variant.sh <name> <flags...>
#   -> 3d-condtional-stable-diffusion_amd/csrc/variants/<name>.so     (run anything with DM3D_LIB=<that path>)
# e.g. tools/mk_variant.sh noguard -DDM3D_NO_RANGE_GUARD
set -e
name=$1; shift
cd "$(dirname "$0")/../3d-condtional-stable-diffusion_amd/csrc"
mkdir -p variants/obj_$name
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -ffp-contract=off "$@" -c $f -o variants/obj_$name/${f%.hip}.o &
  objs="$objs variants/obj_$name/${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so $objs
rm -rf variants/obj_$name
echo built variants/$name.so
