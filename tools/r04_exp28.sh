set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_hip_primitives.py tests/test_full_size_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "potrf or lockstep or factoris or repetitions or logpdf or golden or inducing" > gpurun_out/r04_exp28_tests.txt 2>&1
O=gpurun_out/r04_exp28.txt; : > $O
python tools/run_config.py C4 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
python tools/run_config.py C2 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
python tools/time_small_layers.py 512:4 1024:4 2048:4 2>/dev/null | grep -o "n=[0-9]* p=[0-9]*\|lockstep+lookahead [0-9.]* ms ([-0-9.]*)" | tr '\n' ' ' >> $O; echo >> $O
