#!/bin/bash
# Re-sweep of the panel grouping of gpar_potrf with the second-generation panel kernel: C3 evaluation time against
# GPAR_POTRF_GROUP / GPAR_POTRF_PAIR_ROWS (panels per trailing update while that many rows remain), same session.
cd "$(dirname "$0")/.."
for cfg in "3 9216" "3 8192" "3 7168" "3 6144" "4 9216" "4 8192" "4 6144" "2 9216" "2 6144" "3 9216"; do
    set -- $cfg
    echo "== GROUP=$1 PAIR_ROWS=$2"
    GPAR_POTRF_GROUP=$1 GPAR_POTRF_PAIR_ROWS=$2 timeout 300 python tools/run_config.py C3 --evals 6 --warmup 2 2>&1 | tail -2 | cut -c1-260
done
