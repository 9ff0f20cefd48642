cd "$GRAFT_REPO_ROOT"
run() { python tools/run_config.py $1 --evals 6 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config'], 'ms', [round(x, 2) for x in d['ms']])"; }
run C2 default
GPAR_LAYER_PIPELINE=4 run C2 pipe4
GPAR_LAYER_PIPELINE=2 run C2 pipe2
GPAR_POTRF_LOOKAHEAD=1 run C2 la1
GPAR_POTRF_LOOKAHEAD=1 GPAR_LAYER_PIPELINE=4 run C2 la1pipe4
GPAR_POTRF_LOOKAHEAD=0 run C5 c5_la0
run C5 c5_default
GPAR_LAYER_PIPELINE=4 run C5 c5_pipe4
GPAR_LAYER_PIPELINE=2 run C5 c5_pipe2
GPAR_LAYER_PIPELINE=3 run C3 c3_pipe3
run C3 c3_default
for n in 2048 4096; do GPAR_POTRF_LOOKAHEAD=1 python tools/time_potrf.py $n 2>&1 | grep potrf; GPAR_POTRF_LOOKAHEAD=0 python tools/time_potrf.py $n 2>&1 | grep potrf; done
