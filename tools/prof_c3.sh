# kernel stats of config C3's registration with the coarse modes of tools/time_c3_modes.py -> gpurun_out/prof_c3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c -- python $R/tools/time_c3_modes.py > $O/out.txt 2> $O/err.txt
tail -4 $O/out.txt
python - <<PY
import csv, glob
f = glob.glob("$O/**/c_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
