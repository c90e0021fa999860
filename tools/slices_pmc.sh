cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sl in 18 21 24 29; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    VFM_SLICES=$sl timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/slpmc/s${sl}_$set -o p -- python $R/tools/prof_match.py 3 > /dev/null 2>&1
  done
  python - <<PY
import csv,glob
tot={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    v=[]
    for f in glob.glob("$R/gpurun_out/slpmc/s${sl}_%s/*counter_collection.csv"%c):
        for r in csv.DictReader(open(f)):
            if "match_coarse" in r["Kernel_Name"] and r["Counter_Name"]==c: v.append(float(r["Counter_Value"]))
    tot[c]=sum(v)/max(len(v),1)
print("slices $sl traffic GB", (2*tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024/1e9)
PY
done
cd $R
for sl in 18 21 24 29 21 29; do VFM_SLICES=$sl python bench.py --no-cpu-baseline --no-extra --steps 80 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slices $sl', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['single_stream']['avg_launch_ms'],3))"; done
