#!/bin/bash
# build_lib_variant.sh NAME "-DFLAG ..." -> build/variants/libsplat_one_amd_NAME.so (load with SPLAT_ONE_AMD_LIB=...)
set -e
cd "$(dirname "$0")/../splat_one_amd/csrc"
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast-honor-pragmas $2 -shared \
  common.hip projection.hip sh.hip isect.hip rasterize_fwd.hip rasterize_bwd.hip rasterize_bwd_tile.hip adam.hip loss.hip preprocess.hip raster_op.hip step.hip mcmc.hip refine.hip \
  -o ../../build/variants/libsplat_one_amd_$1.so
