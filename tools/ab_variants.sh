# Dev aid: A/B of several builds of the library (gpar_amd/libv_<name>.so) in one GPU session: bench step (+ optional configs).
cp gpar_amd/libgpar_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp gpar_amd/libv_$v.so gpar_amd/libgpar_hip.so
    python bench.py --no-extras --no-cpu --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$v: ms/step', round(d['ms_per_step'],2), 'live', round(r['achieved'],1), 'iso', round(r['isolated']['achieved'],1))"
    python tools/run_config.py C5 --evals 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   C5', round(sorted(d['ms'])[len(d['ms'])//2], 2))"
  done
done
cp /tmp/keep.so gpar_amd/libgpar_hip.so
