set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r04_exp21.txt; : > $O
for v in 2048 4096 8192; do
  echo "== GPAR_LOCKSTEP_FUSED_BUILD_ROWS=$v" >> $O
  GPAR_LOCKSTEP_FUSED_BUILD_ROWS=$v python tools/run_config.py C2 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
  GPAR_LOCKSTEP_FUSED_BUILD_ROWS=$v python tools/time_small_layers.py 3072:4 4096:8 6144:4 2>/dev/null | grep -o "n=[0-9]* p=[0-9]*\|lockstep+lookahead [0-9.]* ms ([-0-9.]*)" | tr '\n' ' ' >> $O; echo >> $O
done
