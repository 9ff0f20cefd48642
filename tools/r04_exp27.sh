set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=gpurun_out/prof_r04exp27; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C4 --evals 2 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_exp27_C4_timeline.txt 2>&1
rm -rf $D
