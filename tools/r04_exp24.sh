set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_primitives.py tests/test_parity_gpu.py -x -q -m gpu -k "single_column or gemm or golden or posterior or moments" > gpurun_out/r04_exp24_tests.txt 2>&1
O=gpurun_out/r04_exp24.txt; : > $O
for v in "" "GPAR_GEMV=0"; do
  echo "== $v" >> $O
  env $v python tools/run_config.py C4 --evals 7 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done
