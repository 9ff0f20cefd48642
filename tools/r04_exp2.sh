set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/r04_small_profile.py 512 4 > gpurun_out/r04_exp2_host512.txt 2>&1
D=gpurun_out/prof_r04exp2; rm -rf $D
PROFILE_HOST=0 rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/r04_small_profile.py 512 4 > $D.log 2>&1
python tools/eval_timeline.py $D 100 > gpurun_out/r04_exp2_timeline512.txt 2>&1
rm -rf $D
