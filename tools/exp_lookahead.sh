cd "$GRAFT_REPO_ROOT"
run() { python tools/run_config.py $1 --evals 5 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config'], 'ms', [round(x, 2) for x in d['ms']])"; }
for n in 6144 8192 12288 16384; do for la in 0 1; do echo -n "LA=$la "; GPAR_POTRF_LOOKAHEAD=$la python tools/time_potrf.py $n 2>&1 | grep potrf; done; done
GPAR_POTRF_LOOKAHEAD=0 run C3 c3_la0
run C3 c3_default
GPAR_POTRF_LOOKAHEAD=0 GPAR_LAYER_PIPELINE=2 run C5 c5_la0_pipe2
GPAR_POTRF_LOOKAHEAD=0 GPAR_LAYER_PIPELINE=3 run C5 c5_la0_pipe3
GPAR_POTRF_LOOKAHEAD=0 GPAR_LAYER_PIPELINE=4 run C5 c5_la0_pipe4
GPAR_POTRF_LOOKAHEAD=0 GPAR_LAYER_PIPELINE=3 run C3 c3_la0_pipe3
