#!/bin/bash
# LDS counters of the ViT kernels at $1 images (default 90): how busy the LDS is beside the matrix pipes (one --pmc pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_vit_lds
rm -rf $O && mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p$i -- python $R/tools/prof_vit.py 1 3 ${1:-90} > $O/log$i.txt 2>&1
  echo "pass $i ($set): rc=$?"; tail -2 $O/log$i.txt | cut -c1-200
done
python - $O <<'P'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/**/p*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "vit_" not in n: continue
        k = n.split("(")[0][-60:] if "<" not in n else n[n.find("vit_"):n.find(">") + 1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, {c: round(v) for c, v in a.items()})
    if cyc:
        print(f"    cycles {cyc:.0f}: LDS index active / (cycles x 256 units) = {a.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * 256):.3f}, bank conflict cycles share {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
P
