# kernel timeline of gpar_chol_inverse at n = 16384, recursive inversion against the solve of the identity
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 0; do
    D=gpurun_out/prof_inv_$v; rm -rf $D
    GPAR_INVERSE_RECURSIVE=$v rocprofv3 --kernel-trace -f csv -d $D -o kt -- python tools/run_inverse.py 16384 3 > $D.log 2>&1
    echo "== GPAR_INVERSE_RECURSIVE=$v"; grep chol_inverse $D.log
    python tools/timeline_last_eval.py $D 20 | tail -80
done
