set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
GPAR_FIT_THREADS=1 python tools/time_small_fit.py 100:3 400:3 1024:4 2048:4 > gpurun_out/r04_exp12_smallfit_1thread.txt 2>&1
PROFILE=1 python tools/time_small_fit.py 400:3 > gpurun_out/r04_exp12_profile.txt 2>&1
