# kernel stats of the default bench command (pipelined) -> gpurun_out/prof_bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bench
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $R/bench.py --no-cpu-baseline --no-extra $@ > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt | cut -c1-200
python - <<PY
import csv, glob
f = glob.glob("$O/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
