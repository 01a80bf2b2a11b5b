#!/bin/bash
# Rebuild the library, then run a command on the GPU box (a stale .so against new Python bindings shifts arguments:
# round 3 lost a run to that).  A failed build stops here.  usage: tools/gpu.sh [--timeout S] -- '<command>'
HERE="$(dirname "$0")"
make -C "$HERE/../splat_one_amd/csrc" -j8 > /tmp/so_build.log 2>&1 || { grep -n "error" -A6 /tmp/so_build.log | head -60; echo "BUILD FAILED: not going to the GPU"; exit 1; }
make -C "$HERE/../oracle" > /tmp/so_build_oracle.log 2>&1 || { tail -20 /tmp/so_build_oracle.log; echo "ORACLE BUILD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
