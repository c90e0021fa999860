# which hardware queues the pipeline's kernels run on, pipeline after pipeline (same construction sequence as tools/queue_probe.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/queue_probe
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/qp.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/vfm-registration_amd")
import torch
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p) for p in range(2)]
ev = torch.cuda.Event(); ev.record()
keep = []
for it in range(9):
    if it: keep.append(torch.cuda.Stream())
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="int8-half")
    keep.append(pipe.prep_stream); keep.extend(pipe.solve_streams)
    for i in range(6):
        p = pairs[i % 2]; pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
    pipe.synchronize(); torch.cuda.synchronize()
    torch.zeros(1, device="cuda").sum().item()   # marker kernels between pipelines
    del pipe
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python /tmp/qp.py > $O/out.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
it, cur, seen = 0, {}, 0
for r in rows:
    nm = r["Kernel_Name"]
    key = "coarse" if "match_coarse" in nm else "prep" if "prep_chunk" in nm else "select" if "select_half" in nm else "ransac" if "ransac_final" in nm else None
    if key:
        cur.setdefault(key, set()).add(r["Queue_Id"])
        if key == "coarse":
            seen += 1
    if seen == 6 and key == "ransac" and len(cur.get("ransac", ())) >= 1 and sum(1 for x in rows if 0) == 0:
        pass
# split per pipeline: every 6 coarse launches
out, cnt, cur = [], 0, {}
for r in rows:
    nm = r["Kernel_Name"]
    key = "coarse" if "match_coarse" in nm else "prep" if "prep_chunk" in nm else "solve" if ("select_half" in nm or "ransac_final" in nm) else None
    if not key:
        continue
    if key == "coarse":
        if cnt == 6:
            out.append(cur); cur, cnt = {}, 0
        cnt += 1
    cur.setdefault(key, []).append(r["Queue_Id"])
out.append(cur)
for i, c in enumerate(out):
    print(f"pipeline {i}: coarse on queue(s) {sorted(set(c.get('coarse', [])))}, prep on {sorted(set(c.get('prep', [])))}, solve on {sorted(set(c.get('solve', [])))}")
PY
