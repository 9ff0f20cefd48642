set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_primitives.py -x -q -m gpu -k "explicit_inverse or chol_inverse or gemm" > gpurun_out/r04_exp16_tests.txt 2>&1
O=gpurun_out/r04_exp16_c4.txt; : > $O
for v in "" "GPAR_TRSM_INVERSE_SPREAD_MAX=0"; do
  echo "== $v" >> $O
  env $v python tools/run_config.py C4 --evals 7 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done
