#!/bin/bash
# C2 (n = 4096, p = 4): pipeline depth / look-ahead / half tiles after the host-side changes, same session
cd "$(dirname "$0")/.."
run() { python tools/run_config.py C2 --evals 10 --warmup 3 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); ms = sorted(d['ms']); print('$1', 'median', round(ms[len(ms)//2], 2), 'min', round(ms[0], 2))"; }
run default
GPAR_LAYER_PIPELINE=3 run pipe3
GPAR_LAYER_PIPELINE=2 run pipe2
GPAR_POTRF_LOOKAHEAD=1 run lookahead_forced
GPAR_GEMM_HALF_TILES=600 run half600
GPAR_POTRF_NBO=256 run nbo256
run default
