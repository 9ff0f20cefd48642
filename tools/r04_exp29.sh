cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_exp29.txt; : > $O
for v in "" "GPAR_POTRF_LA_SMALL_TILES=0" "GPAR_POTRF_SMALL_UPDATE=0" "GPAR_VFE_FUSED_SCALARS=0" "GPAR_GEMV=0" "GPAR_ONE_CALL_GRAD_ROWS=0" "GPAR_AOT=0" "GPAR_POTRF_FUSED=0" "GPAR_TRSM_FUSED=0"; do
  echo "== $v" >> $O
  env $v python tools/r04_fuzz_one.py 507 516 2>&1 | grep -v amdgpu >> $O
done
