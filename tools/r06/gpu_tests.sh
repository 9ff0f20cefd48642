# The round-end checks on the GPU box: every -m gpu test, then smoke().
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6t; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
