# kernel timeline of the pipelined registration (mode $1, default mx6-half; data $2: d2 (default) or lifted; $3 = False: no overlap, every kernel alone): per HIP stream (queue) the busy time and the
# kernels, per registration -> gpurun_out/trace_pipe
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_pipe_${1:-mx6-half}_${2:-d2}_${3:-True}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pipe_only.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/vfm-registration_amd")
import torch
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
n, m, d = 20000, 200000, 384
pairs = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device="cuda", clouds=10, view_noise=0.1, common=1.0) if "${2:-d2}" == "lifted" else synth.make_pair_device(n, m, d, seed=42 + p) for p in range(2)]
ev = torch.cuda.Event(); ev.record()
pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=${3:-True}, overlap_prepare=${3:-True}, solve_streams=2, coarse="${1:-mx6-half}")
for i in range(60):
    p = pairs[i % 2]; pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
pipe.synchronize(); torch.cuda.synchronize()
print("records", pipe._records(), "rescanned chunks per query", (pipe.last_rescans or 0) / n)
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python /tmp/pipe_only.py > $O/out.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "vfmm" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"] or "hipcub" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 registrations: from the 21st coarse launch on
co = [i for i, r in enumerate(rows) if "match_coarse" in r["Kernel_Name"]]
rows = rows[co[20]:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
nreg = len([r for r in rows if "match_coarse" in r["Kernel_Name"]])
print(f"{nreg} registrations in {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e6 / nreg:.3f} ms each; columns: {list(rows[0].keys())[:14]}")
key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append(r)
for q, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    names = collections.Counter(r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44] for r in rs)
    per = collections.defaultdict(float)
    for r in rs:
        per[r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.2f} ms = {busy / nreg:.3f} ms per registration")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:9]:
        print(f"      {k:46s} {v / nreg:8.1f} us per registration ({names[k] / nreg:.1f} launches)")
print("timeline of two registrations (kernels longer than 8 us; unnamed = RANSAC / torch kernels):")
allrows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
co = [i for i, r in enumerate(allrows) if "match_coarse" in r["Kernel_Name"]]
sub = allrows[co[30]:co[32]]
t00 = int(sub[0]["Start_Timestamp"])
for r in sub:
    nm = r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    a0, a1 = (int(r["Start_Timestamp"]) - t00) / 1e3, (int(r["End_Timestamp"]) - t00) / 1e3
    if a1 - a0 > 8 or "coarse" in nm or "prep" in nm:
        print(f"  queue {r['Queue_Id']} {a0:8.1f} -> {a1:8.1f} us ({a1 - a0:6.1f})  {nm}")
PY
