cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_r02k_C3; rm -rf $D
rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py C3 --evals 3 --warmup 1 > $D.log 2>&1
python tools/kernel_table.py $D "C3 kernel table (4 evaluations, pipelined)" > gpurun_out/r02k_C3_kernel_stats.txt 2>&1
find $D -name "*.db" -delete
