# the multi-GPU bench path with two ranks sharing the one GPU of the box (gloo instead of RCCL)
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_BACKEND=gloo
for mode in ${1:-auto}; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --dp-mode $mode 2>&1 | grep -v "amdgpu.ids\|OMP_NUM\|socket.cpp\|Gloo\|\*\*\*" | cut -c1-1200 | tail -6
done
