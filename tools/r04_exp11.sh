set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "one_call_objective or gradient" > gpurun_out/r04_exp11_tests.txt 2>&1
python tools/time_small_fit.py 100:3 400:3 1024:4 2048:4 > gpurun_out/r04_exp11_smallfit.txt 2>&1
GPAR_ONE_CALL_GRAD_ROWS=0 python tools/time_small_fit.py 100:3 400:3 1024:4 2048:4 > gpurun_out/r04_exp11_smallfit_off.txt 2>&1
