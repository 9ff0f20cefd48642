cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/r06/c2_predict.py
echo "--- GPAR_SPIN_CHAIN=0"; GPAR_SPIN_CHAIN=0 python tools/r06/c2_predict.py
