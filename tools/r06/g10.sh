cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6h; mkdir -p $O
GPAR_BENCH_TRACE=1 python tools/r06/bench_trace.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
grep -A 30 "C2 predict" $O/bench.err | head -50
