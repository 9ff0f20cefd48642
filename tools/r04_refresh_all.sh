set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/refresh_profiles.sh > gpurun_out/r04_refresh.log 2>&1
bash tools/profile_configs.sh r04 C2 C4 C5 > gpurun_out/r04_profile_configs.log 2>&1
bash tools/pmc_mfma_configs.sh > gpurun_out/r04_pmc_mfma.log 2>&1
python bench.py --p 1 --no-extras --no-cpu --steps 10 --warmup 3 > gpurun_out/r04_bench_p1.json 2>gpurun_out/r04_bench_p1.err
rm -rf gpurun_out/prof_r04_C*/ 2>/dev/null
du -sh gpurun_out
