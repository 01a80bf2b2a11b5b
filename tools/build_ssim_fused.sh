#!/bin/bash
# Build tools/probes/ssim_fused_t<THREADS>w<WAVES>g<TAPGROUP>.bin: k_ssim_l1_fused of loss.hip next to the
# forward/backward pair, same inputs, for a few shapes (rows per workgroup are a run-time argument).
set -e
cd "$(dirname "$0")/.."
rm -f tools/probes/ssim_fused_*.bin
for v in ${FUSED_VARIANTS:-512,2,4 512,2,6 256,2,6}; do
  IFS=, read -r t w g <<< "$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -DSO_FUSED_THREADS=$t -DSO_FUSED_WAVES=$w -DSO_FUSED_TAPGROUP=$g \
    -Rpass-analysis=kernel-resource-usage splat_one_amd/csrc/loss.hip splat_one_amd/csrc/common.hip tools/probes/ssim_bench.hip \
    -o tools/probes/ssim_fused_t${t}w${w}g${g}.bin 2> /tmp/ssim_fused_t${t}w${w}g${g}.log &
done
wait
for f in /tmp/ssim_fused_t*.log; do echo $f; grep -h -A12 "Function Name: _ZN2so15k_ssim_l1_fusedILi3" $f | grep -E "VGPRs:|Occupancy|Spill|ScratchSize|LDS Size" | tr '\n' ' ' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/remark://g; s/[a-z_\/.]*hip:[0-9]*:[0-9]*://g'; echo; grep -E "error" -A3 $f | head -10; done
