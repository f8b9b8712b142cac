#!/bin/bash
# Like tools/mk_variant.sh, but recompiles only the 16x16x32 conv kernel files (the ones that include dm3d_conv_h3v2_parts.h) with the extra
# flags and links them with the product's other objects:  tools/mk_variant_conv.sh <name> <flags...>  -> csrc/variants/<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../3d-condtional-stable-diffusion_amd/csrc"
make -s -j8 > /dev/null
mkdir -p variants/obj_$name
objs=""
for f in *.hip; do
  case $f in
    dm3d_conv_h3w.hip|dm3d_conv_h3v3.hip|dm3d_conv_h3_host.hip)
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -ffp-contract=off "$@" -c $f -o variants/obj_$name/${f%.hip}.o &
      objs="$objs variants/obj_$name/${f%.hip}.o" ;;
    *) objs="$objs ${f%.hip}.o" ;;
  esac
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so $objs
rm -rf variants/obj_$name
echo built variants/$name.so
