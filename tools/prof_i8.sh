# kernel breakdown of one search with the int8 dense pass (variant from $1, default 9)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_i8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VFM_AB_VARIANTS=${1:-0} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o i8 -- python $R/tools/ab_i8.py > $O/out.txt 2> $O/err.txt
tail -3 $O/out.txt
python - <<PY
import csv, glob
f = glob.glob("$O/**/i8_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
