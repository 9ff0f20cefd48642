# rocprofv3 kernel statistics of the BASELINE configs (run through gpurun):  bash tools/profile_configs.sh r03 C2 C4 C5
# Summaries land in gpurun_out/<tag>_<config>_kernel_stats.txt; copy them to profiles/.
set -u
TAG=${1:-r03}; shift
CONFIGS=${@:-C2 C4 C5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in $CONFIGS; do
    D=gpurun_out/prof_${TAG}_$c
    rm -rf $D
    python tools/run_config.py $c --evals 5 --warmup 2 > gpurun_out/${TAG}_${c}_timing.json 2> gpurun_out/${TAG}_${c}_timing.err
    python tools/run_config.py $c --evals 5 --warmup 2 --serial > gpurun_out/${TAG}_${c}_timing_serial.json 2>> gpurun_out/${TAG}_${c}_timing.err
    rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/run_config.py $c --evals 3 --warmup 1 > $D.log 2>&1
    {
        echo "rocprofv3 --kernel-trace --stats -- python tools/run_config.py $c --evals 3 --warmup 1   (1x MI355X; 4 evaluations in the trace)"
        echo "timing (not under the profiler): $(cat gpurun_out/${TAG}_${c}_timing.json)"
        echo "timing, serial (no pipelining / look-ahead): $(cat gpurun_out/${TAG}_${c}_timing_serial.json)"
        python tools/kernel_table.py $D "kernel table"
    } > gpurun_out/${TAG}_${c}_kernel_stats.txt 2>&1
    find $D -name "*.db" -delete
done
