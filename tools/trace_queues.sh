# Which hardware queue the pipeline's kernels run on: bench.py plain and under torch.distributed.run (world 1), rocprofv3 kernel trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_queues
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/plain -o t -- python $R/bench.py --no-extra --no-cpu-baseline --steps 40 > $O/plain.json 2> $O/plain.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trun -o t -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 $R/bench.py --gpus 1 --no-extra --no-cpu-baseline --steps 40 > $O/trun.json 2> $O/trun.err
python - <<PY
import csv, glob, collections
for tag in ("plain", "trun"):
    fs = glob.glob("$O/" + tag + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in fs:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    co = [i for i, r in enumerate(rows) if "match_coarse_mx6" in r["Kernel_Name"]]
    rows = rows[co[-30]:]      # the last 30 registrations
    by = collections.defaultdict(collections.Counter)
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "").replace("vfmm::", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][:34]
        by[r["Queue_Id"]][name] += 1
    print("==", tag, "(", len(fs), "trace files )")
    for q, c in sorted(by.items()):
        print("  queue", q, dict(c.most_common(7)))
PY
