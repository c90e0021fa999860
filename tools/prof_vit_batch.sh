# per-kernel times of the ViT forward at 6 and 96 images per call, direct and LDS-tiled GEMMs -> gpurun_out/prof_vit_batch
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vit_batch
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "6 0" "96 0" "96 1"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$1_lds$2 -o b -- python $R/tools/prof_vit.py 1 5 $1 $2 > $O/out_$1_$2.txt 2> $O/err_$1_$2.txt
  echo "== images $1, LDS-tiled GEMMs from $2 workgroups (0 = never)"
  python - <<PY
import csv, glob
f = glob.glob("$O/n$1_lds$2/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
done
