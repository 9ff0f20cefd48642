#!/bin/bash
# Re-sweep of the panel grouping of gpar_potrf at C3 (same session)
cd "$(dirname "$0")/.."
run() { python tools/run_config.py C3 --evals 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ms = sorted(d['ms']); print('$1', 'median', round(ms[len(ms)//2], 2), [round(x, 1) for x in d['ms']])"; }
run default
GPAR_POTRF_PAIR_FIRST=1 run pair_first
GPAR_POTRF_PAIR_FIRST=1 GPAR_POTRF_GROUP=4 run pair_first_group4
run default
GPAR_POTRF_PAIR_FIRST=1 run pair_first
