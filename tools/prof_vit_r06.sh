# per-kernel times of the ViT forward with and without vit_qkv_attention_kernel (round 6) -> gpurun_out/prof_vit_r06
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vit_r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in ${1:-84 90}; do
 for fq in -1 1; do
  VFM_FUSED_QKV=$fq timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n${n}_$fq -o b -- python $R/tools/prof_vit.py 1 6 $n > $O/out_${n}_$fq.txt 2> $O/err_${n}_$fq.txt
  echo "== images $n, vit_fused_qkv $fq"
  python - <<PY
import csv, glob
f = glob.glob("$O/n${n}_$fq/**/b_kernel_stats.csv", recursive=True)
if f:
    tot = 0.0
    for r in list(csv.DictReader(open(f[0])))[:12]:
        if "vit_" in r['Name']:
            tot += float(r['TotalDurationNs']) / 6e3
        print(f"{r['Name'][:86]:86s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
    print(f"sum of the vit kernels per forward: {tot:.1f} us")
PY
 done
done
