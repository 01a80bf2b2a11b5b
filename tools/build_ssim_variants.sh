#!/bin/bash
# Build tools/probes/ssim_t<THREADS>r<ROWS>w<WAVES>.bin for a few tile shapes (travels to the GPU box with the repo).
set -e
cd "$(dirname "$0")/.."
rm -f tools/probes/ssim_t*.bin tools/probes/ssim_[0-9]*.bin
for v in ${SSIM_VARIANTS:-256,36,3 256,36,4 256,27,3}; do
  IFS=, read -r a b c <<< "$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -DSO_SSIM_THREADS=$a -DSO_SSIM_ROWS=$b -DSO_SSIM_WAVES=$c $SSIM_EXTRA \
    -Rpass-analysis=kernel-resource-usage splat_one_amd/csrc/loss.hip splat_one_amd/csrc/common.hip tools/probes/ssim_bench.hip \
    -o tools/probes/ssim_t${a}r${b}w${c}.bin 2> /tmp/ssim_t${a}r${b}w${c}.log &
done
wait
for f in /tmp/ssim_t*.log; do echo $f; grep -h -A12 "Function Name: _ZN2so13k_ssim_l1_...ILi3" $f | grep -E "VGPRs:|Occupancy|Spill|ScratchSize" | tr '\n' ' ' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/remark://g; s/splat_one_amd\/csrc\/loss.hip:[0-9]*:1://g'; echo; grep -E "error" -A3 $f | head -10; done
