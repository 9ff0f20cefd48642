cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6f; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo rc=$?; tail -3 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6f/bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'iso', d['roofline']['isolated']['frac'], 'launches/step', d['roofline']['launches_per_step'], 'wall', d['wall_s'])
g=d['config_grid']
for k in ('C1','C2','C4','C5'):
    r=g[k]; print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in r.items() if a.endswith('_ms') or a.endswith('ms_best') or a in ('fit_evaluations',)})
    c=r.get('cpu_baseline',{}); print('   cpu', {a:(round(b,2) if isinstance(b,float) else b) for a,b in c.items() if a.endswith('_ms') or a=='cores' or a=='error'})
print('lone', g['lone_factorisation_ms']); print('p1', g['p1_ms_per_step'])
for r in d['small_n']['rows']: print({a:(round(b,2) if isinstance(b,float) else b) for a,b in r.items() if 'ms' in a or a=='n'})
print('fit_predict', {a:(round(b,3) if isinstance(b,float) else b) for a,b in d['fit_predict'].items() if a.endswith('_ms') or 'frac' in a})
P
