# One parameterised A/B sweep (replaces the thirty-one tools/r04_expN.sh one-liners of round 4; their results are in profiles/r04_exp_*.txt
# and NOTES.md section 7).  Run through gpurun:
#
#   gpurun --timeout 1500 -- 'bash tools/ab_sweep.sh <tag> "<workloads>" "<env settings of variant 1>" "<variant 2>" ...'
#
#   <workloads>  space-separated names out of: C2 C3 C4 C5 (tools/run_config.py), p1 (bench.py --p 1: the lone n = 16384 layer),
#                potrf:<n>[,<n>...] (tools/time_potrf_quick.py), small:<n>:<p>[,...] (tools/time_small_layers.py), tests:<pytest -k expression>
#   a variant    "" (the defaults) or "GPAR_X=1 GPAR_Y=2"
# Every (variant, workload) pair appends one line to gpurun_out/<tag>.txt: the variant, the workload, the log marginal likelihood
# (same bits or not) and the best time.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; WORK=$2; shift 2
mkdir -p gpurun_out
O=gpurun_out/$TAG.txt; : > $O
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  for w in $WORK; do
    printf '%s | %s | ' "${v:-defaults}" "$w" >> $O
    case $w in
      C2|C3|C4|C5) env $v python tools/run_config.py $w --evals 9 --warmup 2 2>/dev/null | grep -o '"logpdf": [-0-9.e]*\|"ms_best": [0-9.]*' | tr '\n' ' ' >> $O ;;
      p1) env $v python bench.py --p 1 --no-extras --no-cpu --steps 8 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"logpdf": [-0-9.e]*' | tr '\n' ' ' >> $O ;;
      potrf:*) env $v python tools/time_potrf_quick.py $(echo ${w#potrf:} | tr ',' ' ') 2>/dev/null | grep -o 'n=[0-9]*: [0-9.]* ms\|checksum [-0-9.e+]*' | tr '\n' ' ' >> $O ;;
      small:*) env $v python tools/time_small_layers.py $(echo ${w#small:} | tr ',' ' ') 2>/dev/null | grep -o 'n=[0-9]* p=[0-9]*\|lockstep+lookahead [0-9.]* ms ([-0-9.]*)' | tr '\n' ' ' >> $O ;;
      tests:*) env $v timeout 1500 python -m pytest tests -x -q -m gpu -k "${w#tests:}" 2>&1 | tail -1 | tr '\n' ' ' >> $O ;;
      *) printf 'unknown workload' >> $O ;;
    esac
    echo >> $O
  done
done
cat $O
