cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests/test_fastfit.py -m gpu -x -q > $O/test_fastfit.log 2>&1; echo "fastfit rc=$?"; tail -5 $O/test_fastfit.log
timeout 900 python tools/r06/lockstep_fit.py > $O/lockstep_fit.txt 2>&1; cat $O/lockstep_fit.txt
