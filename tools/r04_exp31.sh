python -m pytest tests/test_fuzz_parity_gpu.py -q -x -k "edge_of_definiteness" 2>&1 | tail -5
python -m pytest tests/test_switches_gpu.py tests/test_robustness.py -q -x 2>&1 | tail -3
python tools/fuzz_more.py 500 540 2>&1 | grep -v amdgpu | tail -45
