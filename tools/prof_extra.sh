# kernel stats of bench.py's extra configurations (C3 end to end, C5) -> gpurun_out/prof_extra
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_extra
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o x -- python -c "
import sys, json; sys.path.insert(0, '$R'); sys.path.insert(0, '$R/vfm-registration_amd')
import torch, bench
out = bench.extra_configs(torch.device('cuda', 0))
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.startswith('ms_')} for k, v in out.items() if isinstance(v, dict)}))
" > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt
python - <<PY
import csv, glob
f = glob.glob("$O/**/x_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
