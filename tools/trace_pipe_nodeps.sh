# tools/trace_pipe.sh's timeline of the lifted regime with the preparation's wait for its buffer set's previous user removed (TIMING ONLY:
# the results race) -- does the preparation of pair i + 1 start late because of that dependency, or because no compute unit takes it?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_pipe_nodeps
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pipe_nodeps.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/vfm-registration_amd")
import torch
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
n, m, d = 20000, 200000, 384
pairs = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device="cuda", clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
ev = torch.cuda.Event(); ev.record()
pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="${1:-auto}")
for i in range(60):
    p = pairs[i % 2]
    if i > 20:
        for r in pipe.sets: r.done = None
    pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
pipe.synchronize(); torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python /tmp/pipe_nodeps.py > $O/out.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
co = [i for i, r in enumerate(rows) if "match_coarse" in r["Kernel_Name"]]
t0 = int(rows[co[40]]["Start_Timestamp"])
print("cycle", (int(rows[co[55]]["Start_Timestamp"]) - int(rows[co[35]]["Start_Timestamp"])) / 20e3, "us")
for r in rows[co[40]:co[42]]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s > 8 or "prep" in r["Kernel_Name"]:
        print(f"q{r['Queue_Id']} {s:9.1f} -> {e:9.1f} ({e - s:7.1f}) " + r["Kernel_Name"].replace("vfmm::(anonymous namespace)::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:40])
PY
