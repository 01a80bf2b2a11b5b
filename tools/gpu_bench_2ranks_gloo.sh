# Functional rehearsal of bench.py's N>1 flow on the ONE GPU of the box: two ranks, gloo (RCCL refuses two ranks on one
# device), small sizes.  Numbers mean nothing (gloo stages HIP tensors through the host); the point is that both schemes
# run, the JSON line is complete, and -- with --densify -- the replicas refine on the device and stay in step.
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_BACKEND=gloo
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 20 --warmup 5 --gaussians 20000 --width 640 --height 360 --no-cpu-baseline > gpurun_out/bench2_gloo.json 2> gpurun_out/bench2_gloo.err &&
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
  bench.py --gpus 2 --steps 30 --warmup 5 --gaussians 20000 --width 640 --height 360 --densify 10 --no-cpu-baseline > gpurun_out/bench2_gloo_densify.json 2> gpurun_out/bench2_gloo_densify.err
RC=$?
[ $RC -eq 0 ] && timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --gpus 2 --steps 30 --warmup 5 --gaussians 20000 --width 640 --height 360 --densify 10 --attr-dtype f16 --no-cpu-baseline > gpurun_out/bench2_gloo_densify_f16.json 2> gpurun_out/bench2_gloo_densify_f16.err
echo rc=$?
cut -c1-1500 gpurun_out/bench2_gloo.json gpurun_out/bench2_gloo_densify.json gpurun_out/bench2_gloo_densify_f16.json
grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/bench2_gloo.err gpurun_out/bench2_gloo_densify.err | tail -20
