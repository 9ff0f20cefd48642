set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py -x -q -k "one_call or lockstep" > gpurun_out/r04_exp1_tests.txt 2>&1
python tools/time_small_layers.py 256:4 512:4 1024:4 2048:4 4096:4 > gpurun_out/r04_exp1_small.txt 2>&1
for v in "" "GPAR_POTRF_NBO=256" "GPAR_POTRF_BATCH_REST_AFTER_LA=100000" "GPAR_POTRF_NBO=256 GPAR_POTRF_BATCH_REST_AFTER_LA=100000" "GPAR_ONE_CALL=0"; do
  echo "== $v" >> gpurun_out/r04_exp1_c2.txt
  env $v python tools/run_config.py C2 --evals 9 --warmup 2 >> gpurun_out/r04_exp1_c2.txt 2>&1
done
D=gpurun_out/prof_r04exp1_C2; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C2 --evals 3 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_exp1_C2_timeline.txt 2>&1
rm -rf $D
