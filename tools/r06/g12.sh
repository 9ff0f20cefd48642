cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6j; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); c=d['config_grid']['C2']; print(c['predict_100_samples_ms_all'], c['predict_jit_state'], d['wall_s'])"
