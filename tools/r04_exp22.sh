set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_primitives.py tests/test_full_size_gpu.py -x -q -m gpu -k "potrf or lockstep or factoris or repetitions" > gpurun_out/r04_exp22_tests.txt 2>&1
O=gpurun_out/r04_exp22.txt; : > $O
for v in 512 1024 2048; do
  echo "== GPAR_POTRF_LA_SMALL_TILES=$v" >> $O
  GPAR_POTRF_LA_SMALL_TILES=$v python tools/run_config.py C2 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
  GPAR_POTRF_LA_SMALL_TILES=$v python tools/time_small_layers.py 512:4 1024:4 2048:4 3072:4 2>/dev/null | grep -o "n=[0-9]* p=[0-9]*\|lockstep+lookahead [0-9.]* ms ([-0-9.]*)" | tr '\n' ' ' >> $O; echo >> $O
  GPAR_POTRF_LA_SMALL_TILES=$v python bench.py --p 1 --no-extras --no-cpu --steps 8 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done
D=gpurun_out/prof_r04exp22; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C2 --evals 3 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_exp22_C2_timeline.txt 2>&1
rm -rf $D
