# Dev aid: A/B of two builds of the library in one GPU session.  gpar_amd/libgpar_hip_base.so = the baseline build.
run() {
    python tools/time_gemm.py 16384 512 1024 2>&1 | grep syrk
    python tools/time_gemm.py 4096 512 2>&1 | grep syrk
    python tools/time_gemm.py 2048 512 2>&1 | grep syrk
    python tools/time_potrf.py 4096 8192 16384 2>&1 | grep -E "potrf|gemm NN"
    for mode in single pipelined; do
        if [ $mode = single ]; then export GPAR_LAYER_PIPELINE=1; P=2; else unset GPAR_LAYER_PIPELINE; P=8; fi
        python bench.py --no-extras --no-cpu --steps 8 --warmup 2 --p $P 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$mode p=$P: ms/step', round(d['ms_per_step'],2), 'per layer', round(d['ms_per_step']/$P,2), 'live', round(r['achieved'],1), 'iso', round(r['isolated']['achieved'],1))"
    done
    unset GPAR_LAYER_PIPELINE
}
cp gpar_amd/libgpar_hip.so /tmp/new.so
for rep in 1 2; do
    echo "== new"; cp /tmp/new.so gpar_amd/libgpar_hip.so; run
    echo "== base"; cp gpar_amd/libgpar_hip_base.so gpar_amd/libgpar_hip.so; run
done
cp /tmp/new.so gpar_amd/libgpar_hip.so
