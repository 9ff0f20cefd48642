# Kernel timelines of lone factorisations (run through gpurun): summaries in gpurun_out/r5_potrf_timelines.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl5p; mkdir -p gpurun_out/tl5p
: > gpurun_out/r5_potrf_timelines.txt
for n in ${@:-1024 2048 4096 8192}; do
    rocprofv3 --kernel-trace -d gpurun_out/tl5p/p$n -o kt -- python tools/time_potrf_quick.py $n > gpurun_out/tl5p/p$n.log 2>&1
    D=$(dirname $(find gpurun_out/tl5p/p$n -name "*.db" | head -1))
    echo "== n = $n" >> gpurun_out/r5_potrf_timelines.txt
    python tools/timeline_last_potrf.py $D >> gpurun_out/r5_potrf_timelines.txt 2>&1
done
find gpurun_out/tl5p -name "*.db" -delete
