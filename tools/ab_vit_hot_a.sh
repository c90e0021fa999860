#!/bin/bash
# Timing experiment: the LDS-tiled ViT GEMM with its token operand (A) read from token group 0 by every workgroup -- L2 hits instead of
# the stream from the Infinity Cache / HBM.  Results are wrong by construction; what is compared is the kernel's time.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_vit_hot_a
rm -rf $O; mkdir -p $O
for hot in 0 1; do
  VFM_HOT_A=$hot timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/h$hot -o b -- python $R/tools/prof_vit.py 1 6 ${1:-90} > $O/out_$hot.txt 2> $O/err_$hot.txt
  echo "== hot A = $hot"
  python - $O/h$hot <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/b_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "vit_" in n:
        print(f"{n[:90]:90s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
P
done
