cd $GRAFT_REPO_ROOT
timeout ${2:-600} python $1 2>&1 | grep -v amdgpu.ids | tail -${3:-40}
