# C3 in lock-step: grouping extent / group size / look-ahead, same session (bench step, ms)
run() { python bench.py --steps 6 --warmup 2 --no-extras --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'live', round(r['achieved'],2))"; }
run default
GPAR_POTRF_BATCH_LOOKAHEAD=0 run nolookahead
GPAR_POTRF_PAIR_ROWS=3072 run pair3072
GPAR_POTRF_PAIR_ROWS=9216 run pair9216
GPAR_POTRF_GROUP=4 run group4
GPAR_POTRF_GROUP=2 run group2
GPAR_POTRF_PAIR_FIRST=1 run pairfirst
GPAR_POTRF_BATCH_LOOKAHEAD=0 GPAR_POTRF_GROUP=1 run nolookahead_nogroup
run default
