#!/bin/bash
# Ablation builds for the "single halo-recompute SSIM kernel" question (VERDICT r1 task 5): what do the derivative maps
# cost the two kernels?  base | forward without the map stores | backward without the map loads | 46-row strips (the
# forward work a fused kernel would do per 36 output rows, vertical halo only)
set -e
cd "$(dirname "$0")/.."
rm -f tools/probes/ssimab_*.bin
build() { /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -DSO_SSIM_THREADS=256 -DSO_SSIM_ROWS=$2 -DSO_SSIM_WAVES=3 $3 \
    splat_one_amd/csrc/loss.hip splat_one_amd/csrc/common.hip tools/probes/ssim_bench.hip -o tools/probes/ssimab_$1.bin 2> /tmp/ssimab_$1.log; }
build base 36 "" & build fwd_nostore 36 "-DSO_SSIM_DBG_NOSTORE" & build bwd_nomapload 36 "-DSO_SSIM_DBG_BWD_NOGLOAD" & build rows26 26 "" &
wait
ls -la tools/probes/ssimab_*.bin
