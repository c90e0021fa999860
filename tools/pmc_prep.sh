# HBM traffic of the three forms of the fp6 operand preparation at C2 (flags 24): FETCH_SIZE / WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: FETCH_SIZE doubled on gfx950) -> gpurun_out/pmc_prep/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_prep
rm -rf $O && mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p$i -- python $R/tools/time_prep_forms.py > $O/log$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done
python - <<PY > $O/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in sorted(glob.glob("$O/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "prep_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("void ", "").replace("vfmm::(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$O/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "prep_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].replace("void ", "").replace("vfmm::(anonymous namespace)::", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("operand preparation at C2 (20 000 + 200 000 rows x 384 fp32 = 338 MB in; int8 image + fp6 half image + per-row data out), flags 24;")
print("HBM bytes = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, per launch; duration under the counters")
for k, v in agg.items():
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    d = sorted(dur[k])[len(dur[k]) // 2]
    print(f"{k:40s} FETCH_SIZE {f / 1024:7.1f} MB (x2 = {2 * f / 1024:7.1f})  WRITE_SIZE {w / 1024:7.1f} MB  -> {(2 * f + w) / 1024:7.1f} MB per launch, {d:7.1f} us = {(2 * f + w) * 1024 / d / 1e6:5.2f} TB/s")
PY
cat $O/summary.txt
