# rocprofv3 kernel table + timeline of one BASELINE configuration (run through gpurun): bash tools/r05_config_stats.sh C4
set -u
CFG=${1:-C4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl5/$CFG; mkdir -p gpurun_out/tl5
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/tl5/$CFG -o kt -- python tools/run_config.py $CFG --evals 3 --warmup 1 > gpurun_out/tl5/$CFG.log 2>&1
python tools/kernel_table.py gpurun_out/tl5/$CFG "$CFG" > gpurun_out/r5_${CFG}_kernel_stats.txt 2>&1
(cd tools && python eval_timeline.py ../gpurun_out/tl5/$CFG 400) > gpurun_out/r5_${CFG}_timeline.txt 2>&1
find gpurun_out/tl5 -name "*.db" -delete
