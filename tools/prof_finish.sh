# per-kernel durations of the finish stage (tools/prof_finish.py): $1 record kinds, $2 select variants, $3 data (lifted | d2)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_finish
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/prof_finish.py "${1:-0,5}" "${2:-0}" "${3:-lifted}" > $O/out.txt 2>&1
grep MARK $O/out.txt
python - <<PY
import csv, glob, collections
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
kinds = [int(v) for v in "${1:-0,5}".split(",")]; variants = [int(v) for v in "${2:-0}".split(",")]
co = [i for i, r in enumerate(rows) if "match_coarse" in r["Kernel_Name"]]
k = 0
for rec in kinds:
    for v in variants:
        seg = rows[co[k + 3]:(co[k + 4] if k + 4 < len(co) else len(rows))]   # the 4th repetition
        k += 4
        parts = []
        for r in seg:
            nm = r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            parts.append(f"{nm.replace('match_', '')} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
        print(f"records {rec} variant {v}: " + ", ".join(parts))
PY
