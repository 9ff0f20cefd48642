cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_fastfit.py -m gpu -x -q > gpurun_out/r6a/test_fastfit.log 2>&1; echo "fastfit tests rc=$?"
PROFILE=1 timeout 600 python tools/r06/small_fit.py > gpurun_out/r6a/small_fit.txt 2>&1; echo "small_fit rc=$?"
timeout 600 python tools/r06/launch_table.py 16384 > gpurun_out/r6a/launch_table_16384.txt 2>&1; echo "table rc=$?"
tail -5 gpurun_out/r6a/test_fastfit.log; grep "fit(iters" gpurun_out/r6a/small_fit.txt; grep "==" gpurun_out/r6a/launch_table_16384.txt
