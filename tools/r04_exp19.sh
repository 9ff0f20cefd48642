set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r04_layers_per_rank.txt; : > $O
for p in 8 4 2 1; do
  python bench.py --p $p --no-extras --no-cpu --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('p=$p  ms_per_step %.2f  per layer %.2f  roofline.frac %.3f' % (d['ms_per_step'], d['ms_per_step']/$p, d.get('roofline',{}).get('frac',float('nan'))))" >> $O
done
cat $O
