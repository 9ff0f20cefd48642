# round-4 baseline on this round's box: small-n lock-step timings, C2 timeline, C4 timing + kernel table
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/time_small_layers.py 256:4 512:4 1024:4 2048:4 4096:4 > gpurun_out/r04_small_base.txt 2>&1
python tools/run_config.py C2 --evals 7 --warmup 2 > gpurun_out/r04_C2_base.json 2>gpurun_out/r04_C2_base.err
python tools/run_config.py C4 --evals 5 --warmup 2 > gpurun_out/r04_C4_base.json 2>gpurun_out/r04_C4_base.err
D=gpurun_out/prof_r04base_C2; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C2 --evals 3 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_C2_base_timeline.txt 2>&1
python tools/kernel_table.py $D "C2 baseline" > gpurun_out/r04_C2_base_kernels.txt 2>&1
find $D -name "*.db" -delete
D=gpurun_out/prof_r04base_C4; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C4 --evals 2 --warmup 1 > $D.log 2>&1
python tools/eval_timeline.py $D 300 > gpurun_out/r04_C4_base_timeline.txt 2>&1
find $D -name "*.db" -delete
rm -rf gpurun_out/prof_r04base_C4/*/*.csv.bak 2>/dev/null
ls -la gpurun_out | tail -20
