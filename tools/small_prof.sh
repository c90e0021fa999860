cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 5; do echo "== variant $v"; VFM_VARIANT=$v python $R/tools/time_small.py 2>&1 | grep "fresh"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small -o s -- python $R/tools/time_small.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/small/s_kernel_trace.csv")))
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]) for r in rows)
# find first registration of n=300 in steady state: look for a match_coarse kernel with small grid: print a window of 30 kernels after the 5th coarse
co=[i for i,k in enumerate(ks) if 'match_coarse' in k[2]]
i0=co[5]
t0=ks[i0-3][0]
for k in ks[i0-3:i0+24]:
    print("%8.1f %7.1f %s"%((k[0]-t0)/1e3,(k[1]-k[0])/1e3,k[2]))
PY
