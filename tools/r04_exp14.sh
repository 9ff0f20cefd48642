set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_primitives.py tests/test_full_size_gpu.py tests/test_parity_gpu.py tests/test_fuzz_parity_gpu.py -x -q -m gpu -k "explicit_inverse or inducing or c4 or vfe or sparse" > gpurun_out/r04_exp14_tests.txt 2>&1
O=gpurun_out/r04_exp14_c4.txt; : > $O
for v in "" "GPAR_TRSM_INVERSE_SPREAD_MAX=0" "GPAR_VFE_SPREAD_MAX=1e3"; do
  echo "== $v" >> $O
  env $v python tools/run_config.py C4 --evals 7 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done
