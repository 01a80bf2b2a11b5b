#!/bin/bash
# build/variants/libsplat_one_amd_gp.so = the product library + tools/experiments/rasterize_gp.hip (so_exp_rasterize_bwd_gp),
# loaded by tools/experiments/dbg_gp.py through SPLAT_ONE_AMD_LIB.
set -e
cd "$(dirname "$0")/../splat_one_amd/csrc"
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast-honor-pragmas -shared \
  -Rpass-analysis=kernel-resource-usage \
  common.hip projection.hip sh.hip isect.hip rasterize_fwd.hip rasterize_bwd.hip adam.hip loss.hip preprocess.hip step.hip mcmc.hip refine.hip \
  ../../tools/experiments/rasterize_gp.hip -o ../../build/variants/libsplat_one_amd_gp.so 2> /tmp/gp_build.log || { grep -E "error" -A5 /tmp/gp_build.log | head -40; exit 1; }
grep -A12 "k_rasterize_bwd_gp" /tmp/gp_build.log | grep -E "VGPRs:|SGPRs Spill|Occupancy|ScratchSize|LDS Size" | tr '\n' ' ' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[a-z_\/.]*hip:[0-9]*:[0-9]*: remark://g' | tr -s ' '; echo
