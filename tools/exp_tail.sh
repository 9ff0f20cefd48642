#!/bin/bash
# width of the last panel of gpar_potrf (GPAR_POTRF_TAIL): potrf alone and the configs, same session
cd "$(dirname "$0")/.."
for t in 0 1024 768; do
    echo "== GPAR_POTRF_TAIL=$t"
    GPAR_POTRF_TAIL=$t python tools/time_potrf.py 1024 2048 4096 8192 16384 2>&1 | grep potrf
    for c in C2 C4 C5; do GPAR_POTRF_TAIL=$t python tools/run_config.py $c --evals 6 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  ', d['config'], [round(x, 2) for x in d['ms']])"; done
done
