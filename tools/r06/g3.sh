cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_fastfit.py -m gpu -x -q > $O/test_fastfit.log 2>&1; echo "fastfit rc=$?"; tail -3 $O/test_fastfit.log
timeout 1200 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "c4_full_size" > $O/test_c4.log 2>&1; echo "c4 rc=$?"; tail -15 $O/test_c4.log
timeout 900 python tools/r06/small_fit.py 100 400 1024 2048 > $O/small_fit.txt 2>&1; grep "fast=1" $O/small_fit.txt | grep default
