cd "$GRAFT_REPO_ROOT"
run() { python tools/run_config.py $1 --evals 6 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config'], 'ms', [round(x, 2) for x in d['ms']], repr(d['logpdf']))"; }
for c in C2 C3 C4 C5; do run $c default; done
GPAR_LAYER_PIPELINE=4 run C2 pipe4
GPAR_LAYER_PIPELINE=2 run C2 pipe2
GPAR_LAYER_PIPELINE=4 run C5 pipe4
