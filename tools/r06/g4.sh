cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6d; mkdir -p $O
timeout 600 python tools/time_syrk_shapes.py > $O/syrk.txt 2>&1; cat $O/syrk.txt
timeout 900 python -m pytest tests/test_hip_primitives.py -m gpu -x -q -k "gemm or potrf or syrk" > $O/prim.log 2>&1; tail -3 $O/prim.log
timeout 600 python tools/r06/launch_table.py 16384 > $O/launch_table.txt 2>&1; grep "==" $O/launch_table.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated',{}).get('frac'))"
