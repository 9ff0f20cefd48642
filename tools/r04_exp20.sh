set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_exp20_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_exp20_smoke.txt 2>&1
