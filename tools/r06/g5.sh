cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6e; mkdir -p $O
for n in 4096; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/kt_$n -o kt -- python tools/r06/one_fit.py $n 1 1 2 > $O/kt_$n.log 2>&1
  python tools/kernel_table.py $O/kt_$n "one_fit $n 1" > $O/kstats_$n.txt 2>&1
  rm -rf $O/kt_$n
done
cat $O/kt_4096.log | tail -3; head -30 $O/kstats_4096.txt
python tools/r06/one_fit.py 4096 default 1 3
