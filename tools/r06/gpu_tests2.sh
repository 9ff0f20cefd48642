cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6t; mkdir -p $O
timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_reference_golden.py tests/test_regressor.py tests/test_robustness.py tests/test_share_nothing.py tests/test_switches_gpu.py tests/test_nested_conditioning.py tests/test_oracle.py -q -m gpu -k "not (test_parity_gpu and not one_call_objective and not greedy)" > $O/gpu_tests2.log 2>&1; echo "rc=$?"; tail -8 $O/gpu_tests2.log
