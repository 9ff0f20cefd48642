# Dev aid: panel-kernel grid size against the pipelined (default) C3 bench
for g in 256 192 128 64; do
    echo "== GRID=$g default pipelining p=8"
    GPAR_PANEL_GRID=$g python bench.py --no-extras --no-cpu --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('ms/step', round(d['ms_per_step'],2), 'live', round(r['achieved'],1), 'iso', round(r['isolated']['achieved'],1), 'conc', round(r['concurrency'],2))"
done
