set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_exp7_tests.txt 2>&1
python tools/time_c2_predict.py > gpurun_out/r04_exp7_c2_predict.txt 2>&1
