set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for g in 2 4; do
  GPAR_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus $g --rows 2048 --p 8 --steps 3 --warmup 1 --no-cpu > gpurun_out/r04_exp23_gpus$g.json 2> gpurun_out/r04_exp23_gpus$g.err
  echo "rc=$?" >> gpurun_out/r04_exp23_gpus$g.err
done
# NCCL single-rank communicator path as the driver launches it
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --rows 2048 --p 4 --steps 2 --warmup 1 --no-cpu > gpurun_out/r04_exp23_torchrun1.json 2> gpurun_out/r04_exp23_torchrun1.err
echo "rc=$?" >> gpurun_out/r04_exp23_torchrun1.err
