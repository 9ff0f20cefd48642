# Matrix-pipe busy counters per kernel for the bench command and the BASELINE configs (run through gpurun).  One counter pass
# each (--pmc with --kernel-trace only); kernels are serialised under counter collection: figures for each kernel alone.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_cfg; rm -rf $O; mkdir -p $O
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --kernel-trace -f csv -d $O/bench -o pmc -- python bench.py --steps 1 --warmup 0 --no-extras --no-cpu > $O/bench.log 2>&1
for c in C2 C4 C5; do
    rocprofv3 --pmc $C --kernel-trace -f csv -d $O/$c -o pmc -- python tools/run_config.py $c --evals 1 --warmup 1 > $O/$c.log 2>&1
done
python tools/pmc_mfma_configs.py $O > $O/summary.json 2> $O/summary.err
find $O -name "*.db" -delete
head -c 1500 $O/summary.json
