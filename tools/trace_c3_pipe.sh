# kernel timeline of C3 as a pipeline (EndToEndPipeline: feature stage of pair i + 1 beside the registration of pair i): per hardware queue the
# busy time per pair, and where the ViT's kernels sit relative to the coarse kernel -> gpurun_out/trace_c3_pipe
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_c3_pipe
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/time_c3_pipe.py 0 > $O/out.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", "").split("(")[0][:40]
co = [i for i, r in enumerate(rows) if "match_coarse" in r["Kernel_Name"]]
sub = rows[co[-20]:co[-2]]
t0, t1 = int(sub[0]["Start_Timestamp"]), int(rows[co[-2]]["Start_Timestamp"])
print(f"18 pairs in {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e6 / 18:.3f} ms each")
by = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in sub:
    by[r["Queue_Id"]][nm(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[r["Queue_Id"]][nm(r)] += 1
for q, d in by.items():
    print(f"queue {q}: busy {sum(d.values()) / 18:.1f} us per pair")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:7]:
        print(f"      {k:42s} {v / 18:8.1f} us per pair ({cnt[q][k] / 18:.1f} launches, {v / cnt[q][k]:.1f} us each)")
print("timeline of two pairs (vit kernels as runs per queue):")
sub = rows[co[-6]:co[-4]]
t00 = int(sub[0]["Start_Timestamp"])
run = None
def flush():
    global run
    if run:
        print(f"  queue {run[0]} {run[1]:8.1f} -> {run[2]:8.1f} us ({run[2] - run[1]:6.1f})  {run[4]} x vit kernels, busy {run[3]:.1f} us")
        run = None
for r in sub:
    a0, a1 = (int(r["Start_Timestamp"]) - t00) / 1e3, (int(r["End_Timestamp"]) - t00) / 1e3
    n = nm(r)
    if "vit_" in n or "lift" in n:
        if run and run[0] == r["Queue_Id"] and a0 - run[2] < 40:
            run[2] = a1; run[3] += a1 - a0; run[4] += 1
        else:
            flush(); run = [r["Queue_Id"], a0, a1, a1 - a0, 1]
        continue
    if a1 - a0 > 15 or "coarse" in n or "prep" in n:
        flush()
        print(f"  queue {r['Queue_Id']} {a0:8.1f} -> {a1:8.1f} us ({a1 - a0:6.1f})  {n}")
flush()
PY
