# PMC passes for the kernels of one gated search (coarse + finish stage) on bench.py's lifted descriptors, record kind $1 (default 5):
# instructions per wave, wave lifetime (SQ_WAVE_CYCLES counts quad-cycles), wait shares -> gpurun_out/pmc_finish/summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_finish
rm -rf $O && mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_GDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p$i -- python $R/tools/prof_finish.py "${1:-5}" 0 "${2:-lifted}" > $O/log$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def short(n):
    n = n.replace("vfmm::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48] if ("match_" in n or "prep_" in n or "ransac" in n or "threshold" in n) else None
for f in sorted(glob.glob("$O/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$O/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, cs in agg.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    d = sorted(dur[k])[len(dur[k]) // 2]
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    w = max(a.get("SQ_WAVES", 1), 1)
    out[k] = {"median_us": d, "waves": w, "wave_lifetime_us": 4 * a["SQ_WAVE_CYCLES"] / w / (cyc / d),
              "wave_time_shares": {c: a[c] / a["SQ_WAVE_CYCLES"] for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
              "insts_per_wave": {c[9:]: a[c] / w for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM") if c in a}}
json.dump(out, open("$O/summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["median_us"]):
    ipw = v["insts_per_wave"]
    print(f"{k:48s} {v['median_us']:8.1f} us waves {v['waves']:8.0f} life {v['wave_lifetime_us']:7.1f} us  wait {v['wave_time_shares']['SQ_WAIT_ANY']:.2f} waitinst {v['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f} active {v['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f}  " + " ".join(f"{c} {x:.0f}" for c, x in ipw.items()))
PY
