# Everything the round's record is read from, in one gpurun call: PMC traffic + bench line + kernel stats (tools/refresh_profiles.sh),
# matrix-pipe counters per kernel (tools/pmc_mfma_configs.sh), Gram timings, the launch table of the lone factorisation, small-n fits.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_ARGS="--steps 20 --warmup 5" bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
bash tools/pmc_mfma_configs.sh > gpurun_out/pmc_cfg.log 2>&1
python tools/time_gram_configs.py > gpurun_out/refresh/gram_configs.jsonl 2> gpurun_out/refresh/gram_configs.err
python tools/r06/launch_table.py 16384 > gpurun_out/refresh/launch_table_16384.txt 2>&1
python tools/r06/small_fit.py 100 400 1024 2048 > gpurun_out/refresh/small_fit.txt 2>&1
tail -2 gpurun_out/refresh.log; head -c 600 gpurun_out/refresh/pmc_traffic.json; tail -c 300 gpurun_out/refresh/bench_n1.json
