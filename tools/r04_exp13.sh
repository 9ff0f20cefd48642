set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r04_exp13_fuse_la.txt; : > $O
for v in 0 1100 1600 2100 2600 3100 4200; do
  echo "== GPAR_POTRF_FUSE_LA_ROWS=$v" >> $O
  GPAR_POTRF_FUSE_LA_ROWS=$v python tools/run_config.py C2 --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
  GPAR_POTRF_FUSE_LA_ROWS=$v python tools/time_small_layers.py 1024:4 2048:4 3072:4 2>/dev/null | grep -o "n=[0-9]* p=[0-9]*\|lockstep+lookahead [0-9.]* ms ([-0-9.]*)" | tr '\n' ' ' >> $O; echo >> $O
done
for v in 0 2100 3100 4200 6200; do
  echo "== lone / C5 GPAR_POTRF_FUSE_LA_ROWS=$v" >> $O
  GPAR_POTRF_FUSE_LA_ROWS=$v python bench.py --p 1 --no-extras --no-cpu --steps 8 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"logpdf": [-0-9.e]*' | tr '\n' ' ' >> $O; echo >> $O
  GPAR_POTRF_FUSE_LA_ROWS=$v python tools/run_config.py C5 --evals 5 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O; echo >> $O
done
