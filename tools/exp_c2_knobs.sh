#!/bin/bash
# C2 (n = 4096, p = 4; four factorisations in flight): tile / panel-width knobs, same session
cd "$(dirname "$0")/.."
run() { python tools/run_config.py $CFG --evals 8 $EXTRA 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $EXTRA $1', 'ms', [round(x, 2) for x in d['ms']])"; }
for CFG in C2 C5; do
run default
GPAR_POTRF_NBO=1024 run nbo1024
GPAR_POTRF_NBO=768 run nbo768
EXTRA=--serial run default
EXTRA=--serial GPAR_POTRF_NBO=1024 run nbo1024
EXTRA=
done
python tools/time_potrf.py 2048 4096 8192 2>&1 | grep potrf
GPAR_POTRF_NBO=1024 python tools/time_potrf.py 2048 4096 8192 2>&1 | grep potrf
