# per-kernel times of the ViT forward at 6, 48, 90 and 96 images per call (the library's default kernel policy) -> gpurun_out/prof_vit_r05
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vit_r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 6 48 90 96; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -o b -- python $R/tools/prof_vit.py 1 6 $n > $O/out_$n.txt 2> $O/err_$n.txt
  echo "== images $n"
  python - <<PY
import csv, glob
f = glob.glob("$O/n$n/**/b_kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in list(csv.DictReader(open(f)))[:12]:
    if "vit_" in r['Name']:
        tot += float(r['TotalDurationNs']) / 6e3
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
print(f"sum of the vit kernels per forward: {tot:.1f} us")
PY
done
