cd $GRAFT_REPO_ROOT
for b in tools/probes/ssim_*.bin; do
  echo "== $b"
  timeout 120 $b 2>&1 | grep -v amdgpu.ids
done
