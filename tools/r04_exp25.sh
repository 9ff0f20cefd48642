set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r04_exp25.txt; : > $O
for rep in 1 2 3; do for v in "GPAR_GEMV=1" "GPAR_GEMV=0"; do
  echo -n "$v " >> $O
  env $v python tools/run_config.py C4 --evals 9 --warmup 2 2>/dev/null | grep -o '"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done; done
D=gpurun_out/prof_r04exp25; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C4 --evals 2 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_exp25_C4_timeline.txt 2>&1
rm -rf $D
