set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_switches_gpu.py -x -q -m gpu -k "one_call or lockstep or switch" > gpurun_out/r04_exp8_tests.txt 2>&1
python tools/time_small_layers.py 256:4 512:4 1024:4 2048:4 4096:4 512:8 > gpurun_out/r04_exp8_small.txt 2>&1
python tools/r04_small_profile.py 512 4 > gpurun_out/r04_exp8_host512.txt 2>&1
D=gpurun_out/prof_r04exp8; rm -rf $D
PROFILE_HOST=0 rocprofv3 --kernel-trace --stats -f csv -d $D -o kt -- python tools/r04_small_profile.py 512 4 > $D.log 2>&1
python tools/eval_timeline.py $D 100 > gpurun_out/r04_exp8_timeline512.txt 2>&1
rm -rf $D
