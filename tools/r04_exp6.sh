set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_parity_gpu.py -x -q -m gpu -k "linear_output or sampling or sample" > gpurun_out/r04_exp6_tests.txt 2>&1
python tools/time_c2_predict.py > gpurun_out/r04_exp6_c2_predict.txt 2>&1
python tools/time_c2_predict.py 1000 3 50 500 >> gpurun_out/r04_exp6_c2_predict.txt 2>&1
