# Dev aid: A/B of one environment switch on the single-stream (one layer per GPU, as on 8 GPUs) and pipelined C3 logpdf
# usage: bash tools/ab_env.sh VAR v0 v1
V=$1; shift
for val in "$@" "$@"; do
    for mode in single pipelined; do
        if [ $mode = single ]; then export GPAR_LAYER_PIPELINE=1; P=2; else unset GPAR_LAYER_PIPELINE; P=8; fi
        env $V=$val python bench.py --no-extras --no-cpu --steps 8 --warmup 2 --p $P 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$V=$val $mode p=$P: ms/step', round(d['ms_per_step'],2), 'per layer', round(d['ms_per_step']/$P,2), 'live', round(r['achieved'],1), 'iso', round(r['isolated']['achieved'],1))"
    done
done
