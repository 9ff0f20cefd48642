run() { python bench.py --steps 8 --warmup 2 --no-extras --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'live', round(r['achieved'],2))"; }
for rep in 1 2 3; do
run g3
GPAR_POTRF_GROUP=4 run g4
GPAR_POTRF_GROUP=4 GPAR_POTRF_PAIR_ROWS=4096 run g4_pair4096
GPAR_POTRF_GROUP=5 GPAR_POTRF_PAIR_ROWS=5120 run g5
done
