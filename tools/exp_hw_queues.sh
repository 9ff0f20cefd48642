cd "$GRAFT_REPO_ROOT"
run() { python tools/run_config.py $1 --evals 6 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config'], 'ms', [round(x, 2) for x in d['ms']])"; }
for q in 4 8 16; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q"
run C2 default
GPAR_LAYER_PIPELINE=4 run C2 pipe4
run C5 default
GPAR_LAYER_PIPELINE=4 run C5 pipe4
GPAR_POTRF_LOOKAHEAD=1 run C5 la1
GPAR_POTRF_LOOKAHEAD=1 GPAR_LAYER_PIPELINE=4 run C5 la1_pipe4
run C3 default
GPAR_LAYER_PIPELINE=3 GPAR_POTRF_LOOKAHEAD=1 run C3 pipe3_la1
run C4 default
done
