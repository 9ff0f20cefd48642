#!/bin/bash
# Re-sweep of the panel grouping of gpar_potrf and the pipeline depth at C3 (same session)
cd "$(dirname "$0")/.."
run() { python tools/run_config.py C3 --evals 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ms = sorted(d['ms']); print('$1', 'median', round(ms[len(ms)//2], 2), [round(x, 1) for x in d['ms']])"; }
run default
GPAR_POTRF_PAIR_ROWS=6144 run pair_rows=6144
GPAR_POTRF_PAIR_ROWS=7168 run pair_rows=7168
GPAR_POTRF_PAIR_ROWS=5120 run pair_rows=5120
GPAR_POTRF_GROUP=4 GPAR_POTRF_PAIR_ROWS=6144 run group4_6144
GPAR_LAYER_PIPELINE=3 run pipe3
GPAR_LAYER_PIPELINE=3 GPAR_POTRF_PAIR_ROWS=6144 run pipe3_6144
run default
