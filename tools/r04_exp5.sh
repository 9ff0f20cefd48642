set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# small fits: archive on / off (first process = cold)
python tools/time_small_fit.py 400:3 1024:4 2048:4 3000:4 > gpurun_out/r04_exp5_smallfit_aot.txt 2>&1
GPAR_AOT=0 python tools/time_small_fit.py 400:3 1024:4 2048:4 3000:4 > gpurun_out/r04_exp5_smallfit_noaot.txt 2>&1
# first fit of a process at C3, with and without the archive
python tools/time_fit_c3.py > gpurun_out/r04_exp5_fit_c3_aot.txt 2>&1
GPAR_AOT=0 python tools/time_fit_c3.py > gpurun_out/r04_exp5_fit_c3_noaot.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_exp5_tests.txt 2>&1
