# kernel breakdown of vfm_match_mutual_pairs at C2 size
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pairs
rm -rf $O; mkdir -p $O
true
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pairs_only.py <<PY
import sys
sys.path.insert(0, "$R/vfm-registration_amd")
import torch
from vfmreg import ops, synth
import os
p = synth.make_pair_device(20000, 200000, int(os.environ.get("PAIRS_D", "384")), seed=42)
for _ in range(5):
    ops.match_mutual_pairs(p["q_desc"], p["b_desc"])
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o pairs -- python /tmp/pairs_only.py > $O/out.txt 2> $O/err.txt
python - <<PY
import csv, glob
f = glob.glob("$O/**/pairs_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
