# Kernel timelines of a lone n = 4096 / 2048 factorisation and of the C2 evaluation (run through gpurun); summaries in gpurun_out/tl_*.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
for n in 4096 2048; do
    rocprofv3 --kernel-trace -d gpurun_out/tl/p$n -o kt -- python tools/time_potrf_quick.py $n > gpurun_out/tl/p$n.log 2>&1
    D=$(dirname $(find gpurun_out/tl/p$n -name "*.db" | head -1))
    python tools/timeline_last_potrf.py $D > gpurun_out/tl_potrf_$n.txt 2>&1
done
rocprofv3 --kernel-trace -f csv -d gpurun_out/tl/c2 -o kt -- python tools/run_config.py C2 --evals 3 --warmup 1 > gpurun_out/tl/c2.log 2>&1
(cd tools && python eval_timeline.py ../gpurun_out/tl/c2 300) > gpurun_out/tl_c2.txt 2>&1
find gpurun_out/tl -name "*.db" -delete
