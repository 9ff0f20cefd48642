set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PROFILE=1 python tools/time_small_fit.py 400:3 > gpurun_out/r04_exp10_smallfit_profile.txt 2>&1
