# Kernel timeline + kernel table of the C2 evaluation (run through gpurun); summaries in gpurun_out/r5_c2_*.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl5; mkdir -p gpurun_out/tl5
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/tl5/c2 -o kt -- python tools/run_config.py C2 --evals 3 --warmup 1 > gpurun_out/tl5/c2.log 2>&1
python tools/kernel_table.py gpurun_out/tl5/c2 "C2" > gpurun_out/r5_c2_kernel_stats.txt 2>&1
(cd tools && python eval_timeline.py ../gpurun_out/tl5/c2 300) > gpurun_out/r5_c2_timeline.txt 2>&1
find gpurun_out/tl5 -name "*.db" -delete
