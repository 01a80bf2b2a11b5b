#!/bin/bash
# Rebuild the library, then run a command on the GPU box (a stale .so against new Python bindings shifts arguments:
# round 3 lost a run to that).  usage: tools/gpu.sh [--timeout S] -- '<command>'
set -e
make -C "$(dirname "$0")/../splat_one_amd/csrc" -j8 2>&1 | grep -v "^/opt/rocm\|Entering\|Leaving\|Nothing to be done" || true
make -C "$(dirname "$0")/../oracle" 2>&1 | grep -v "Entering\|Leaving\|Nothing to be done" || true
exec /usr/local/graft/bin/gpurun "$@"
