#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra compiler flags>"
# Kernel experiments: builds rt_kernels.hip again with the given flags and links it with the other objects of the
# in-tree build into build_variants/librt_amd_<name>.so (select it with RT_AMD_LIBRARY).  Not part of the product build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/ray-tracer_amd/csrc
OUT=$ROOT/build_variants
mkdir -p $OUT
make -s -C $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -I$SRC $2 -c $SRC/rt_kernels.hip -o $OUT/rt_kernels_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librt_amd_$1.so $OUT/rt_kernels_$1.o $SRC/_obj/rt_api.o $SRC/_obj/bvh_build.o $SRC/_obj/bvh_device.o $SRC/_obj/rt_comm.o -ldl
rm -f $OUT/rt_kernels_$1.o
echo built $OUT/librt_amd_$1.so
