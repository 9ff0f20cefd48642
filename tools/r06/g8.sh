cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/r06/grid_c1c2.py C1,C2 cpu 2>&1 | tail -3
