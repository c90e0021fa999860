# kernel-trace profile of the bench (pipelined + serial) for the round-2 kernels; copies summaries to gpurun_out/r02prof
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in ${VARIANTS:-0 4}; do
  VFM_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$V -o bench -- python $R/bench.py --no-cpu-baseline --steps 30 > $O/bench_v$V.json 2> $O/prof_v$V.err
  VFM_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$V -o bench -- python $R/bench.py --no-cpu-baseline --steps 30 --streams 1 > $O/bench_s$V.json 2> $O/prof_s$V.err
  for k in v s; do
    f=$(find $O/$k$V -name "*kernel_stats.csv" | head -1)
    echo "== variant $V $k"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
    python -c "import json,sys; d=json.loads(open('$O/bench_$k$V.json').read().strip().splitlines()[-1]); print('value',d['value'],'ms',d['ms_per_step'],'coarse',d['roofline']['avg_launch_ms'])"
  done
done
