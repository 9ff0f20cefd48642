#!/bin/bash
# GPAR_POTRF_PAIR_ROWS (rows left below which panels are no longer grouped): potrf alone, C5, fit-sized inverse path unaffected
cd "$(dirname "$0")/.."
for v in 9216 6144; do
    echo "== GPAR_POTRF_PAIR_ROWS=$v"
    GPAR_POTRF_PAIR_ROWS=$v python tools/time_potrf.py 8192 12288 16384 2>&1 | grep potrf
    GPAR_POTRF_PAIR_ROWS=$v python tools/run_config.py C5 --evals 6 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   C5', [round(x, 2) for x in d['ms']])"
    GPAR_POTRF_PAIR_ROWS=$v GPAR_LAYER_PIPELINE=1 python tools/run_config.py C3 --evals 4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   C3 one stream (as one layer per GPU)', [round(x, 2) for x in d['ms']])"
done
