# MFMA busy counters of the trailing-update kernel (run through gpurun): one counter pass of the bench command (kernels are
# serialised under counter collection, so this is the kernel alone) and one of the bare SYRK shape.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_mfma; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/bench -o pmc -- python bench.py --steps 1 --warmup 0 --no-extras --no-cpu > $O/bench.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace -f csv -d $O/util -o pmc -- python bench.py --steps 1 --warmup 0 --no-extras --no-cpu > $O/util.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/syrk -o pmc -- python tools/time_gemm.py 16384 512 1024 > $O/syrk.log 2>&1
ls -R $O | head -30; tail -3 $O/bench.log $O/util.log $O/syrk.log
