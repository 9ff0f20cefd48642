set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04_exp3_tests.txt 2>&1
python tools/time_small_layers.py 256:4 512:4 1024:4 2048:4 4096:4 > gpurun_out/r04_exp3_small.txt 2>&1
python tools/run_config.py C2 --evals 9 --warmup 2 > gpurun_out/r04_exp3_c2.txt 2>&1
