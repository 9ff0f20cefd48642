# Regenerates the judged artefacts on the GPU box (run through gpurun): bench line, kernel stats of the same command,
# PMC traffic passes.  Outputs land in gpurun_out/refresh/; copy what changed into profiles/ (names: r06_*).
#   PMC first: the bench line then finds the traffic record (stamped with the hash of gemm_f64.h + potrf.h + panel2.h) in profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -f csv -d $O/pmc_$c -o pmc -- python bench.py --steps 1 --warmup 0 --no-extras --no-cpu --no-isolated > $O/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json 2>$O/pmc_traffic.err
# (the bench run below reads the record from profiles/: on the GPU box the copy lives only for this call; commit it from gpurun_out/)
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/r06_bench_pmc_traffic.json
python bench.py ${BENCH_ARGS:-} > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -f csv -d $O/kt -o kt -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_union.py "$T" > $O/union.txt 2>&1
python tools/kernel_table.py $O/kt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu" > $O/kernel_stats.txt 2>&1
rm -rf $O/kt/*/*.db $O/pmc_FETCH_SIZE/*/*.db $O/pmc_WRITE_SIZE/*/*.db
du -sh $O
