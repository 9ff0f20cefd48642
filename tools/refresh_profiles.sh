# Regenerates the judged artefacts on the GPU box (run through gpurun): bench line, kernel stats of the same command,
# PMC traffic passes.  Outputs land in gpurun_out/refresh/; copy what changed into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -f csv -d $O/kt -o kt -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_union.py "$T" > $O/union.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -f csv -d $O/pmc_$c -o pmc -- python bench.py --steps 1 --warmup 0 --no-extras --no-cpu > $O/pmc_$c.log 2>&1
done
rm -rf $O/kt/*/*.db
ISO=$(python -c "import json; print(json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])['roofline']['isolated']['launches'])")
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $ISO > $O/pmc_traffic.json 2>$O/pmc_traffic.err
python tools/kernel_table.py $O/kt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu" > $O/kernel_stats.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE/*/*.db $O/pmc_WRITE_SIZE/*/*.db
du -sh $O
