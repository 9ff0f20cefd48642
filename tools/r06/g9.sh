cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6g; mkdir -p $O
for sc in 1 0; do
GPAR_SPIN_CHAIN=$sc python bench.py --steps 2 --warmup 1 --no-cpu > $O/bench_sc$sc.json 2> $O/bench_sc$sc.err
python -c "
import json; d=json.loads(open('$O/bench_sc$sc.json').read().strip().splitlines()[-1]); c=d['config_grid']['C2']; print('spin_chain=$sc', c['predict_100_samples_ms'], c['fit_20_iters_ms_all'], d['config_grid']['C4'].get('predict_ms'), d['wall_s'])"
done
