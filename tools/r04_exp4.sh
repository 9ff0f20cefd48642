set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_primitives.py tests/test_full_size_gpu.py -x -q -m gpu -k "predicated or inducing or trsm" > gpurun_out/r04_exp4_tests.txt 2>&1
python tools/run_config.py C4 --evals 7 --warmup 2 > gpurun_out/r04_exp4_c4.txt 2>&1
GPAR_VFE_SPREAD_MAX=0 python tools/run_config.py C4 --evals 7 --warmup 2 >> gpurun_out/r04_exp4_c4.txt 2>&1
D=gpurun_out/prof_r04exp4_C4; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C4 --evals 2 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_exp4_C4_timeline.txt 2>&1
python tools/kernel_table.py $D "C4 product-first" > gpurun_out/r04_exp4_C4_kernels.txt 2>&1
rm -rf $D
