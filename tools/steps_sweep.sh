# throughput of the default bench command against the number of timed steps / warm-up steps (the driver runs
# --steps 20 --warmup 5) -> gpurun_out/steps_sweep.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/steps_sweep.txt
: > $O
for KW in "20 5" "20 50" "20 200" "20 5" "200 5" "1000 5"; do
  set -- $KW
  echo "## --steps $1 --warmup $2" >> $O
  VFM_BENCH_TRACE=1 python $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra 2> /tmp/err.txt | cut -c1-200 >> $O
  grep -E "trace|rank" /tmp/err.txt | cut -c1-400 >> $O
done
cat $O
