# per-kernel times of the ViT forward with the token-stationary QKV / fc1 kernel on and off -> gpurun_out/prof_vit_astat
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vit_astat
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in ${@:-"88 0" "88 1" "96 0" "96 1"}; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$1_a$2 -o b -- python $R/tools/prof_vit.py 1 5 $1 -1 $2 > $O/out_$1_$2.txt 2> $O/err_$1_$2.txt
  echo "== images $1, token-stationary kernel from $2 groups (0 = never)"
  python - <<PY
import csv, glob
f = glob.glob("$O/n$1_a$2/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
done
