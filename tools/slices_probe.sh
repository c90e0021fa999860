cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/slices
rm -rf $O; mkdir -p $O
for s in 0 13 16 26 39 55 81; do
  VFM_SLICES=$s timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o f$s -- python $R/tools/prof_match.py 4 > $O/log$s.txt 2>&1
  VFM_SLICES=$s timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o w$s -- python $R/tools/prof_match.py 4 > $O/logw$s.txt 2>&1
done
python - <<PY
import csv
for s in (0, 13, 16, 26, 39, 55, 81):
    def avg(fn, name):
        v=[float(r["Counter_Value"]) for r in csv.DictReader(open(fn)) if "match_coarse" in r["Kernel_Name"] and r["Counter_Name"]==name]
        return sum(v)/len(v)
    f=avg("$O/f%d_counter_collection.csv"%s,"FETCH_SIZE"); w=avg("$O/w%d_counter_collection.csv"%s,"WRITE_SIZE")
    d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open("$O/f%d_kernel_trace.csv"%s)) if "match_coarse" in r["Kernel_Name"])
    print("slices %3d: FETCH %.0f MB (x2 = %.2f GB) WRITE %.0f MB  -> traffic %.2f GB; coarse median %.0f us (under pmc)" % (s, f/1024, 2*f/1024/1024, w/1024, (2*f+w)/1024/1024, d[len(d)//2]))
PY
