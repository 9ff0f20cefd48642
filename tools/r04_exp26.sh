set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_parity_gpu.py tests/test_full_size_gpu.py tests/test_regressor.py tests/test_nested_conditioning.py -x -q -m gpu -k "inducing or sparse or vfe or fitc or dtc or golden or c4 or fuzz or nested" > gpurun_out/r04_exp26_tests.txt 2>&1
O=gpurun_out/r04_exp26.txt; : > $O
for rep in 1 2; do for v in "GPAR_VFE_FUSED_SCALARS=1" "GPAR_VFE_FUSED_SCALARS=0"; do
  echo -n "$v " >> $O
  env $v python tools/run_config.py C4 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done; done
python tools/time_mid_sparse.py >> $O 2>&1
