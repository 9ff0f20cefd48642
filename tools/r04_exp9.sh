set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "nan_pattern or one_call or missing" > gpurun_out/r04_exp9_tests.txt 2>&1
python tools/time_small_layers.py 256:4 512:4 1024:4 2048:4 512:8 > gpurun_out/r04_exp9_small.txt 2>&1
python tools/run_config.py C2 --evals 9 --warmup 2 > gpurun_out/r04_exp9_c2.txt 2>&1
