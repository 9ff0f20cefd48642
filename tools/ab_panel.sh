# A/B of the two panel-kernel generations in one GPU session (run through gpurun):  bash tools/ab_panel.sh
cd "$GRAFT_REPO_ROOT"
for v in 2 1; do
    export GPAR_PANEL_V=$v
    echo "== GPAR_PANEL_V=$v"
    python tools/time_potrf.py 1024 4096 8192 16384 2>&1 | grep potrf
    python tools/time_inverse.py 2>&1 | tail -3
    for c in C2 C3 C4 C5; do python tools/run_config.py $c --evals 4 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config'], 'ms', [round(x, 2) for x in d['ms']], 'logpdf', repr(d['logpdf']))"; done
done
