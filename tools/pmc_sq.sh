# SQ counters of the coarse kernel for a few builds/variants (clock, MFMA busy, wait shares)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc16
mkdir -p $O
for cfg in "pipe::0" "old::1" "skel:libvfmreg_hip_SKEL.so:1"; do
  name=${cfg%%:*}; rest=${cfg#*:}; lib=${rest%%:*}; var=${rest#*:}
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    VFM_LIB=$lib VFM_VARIANT=$var timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o ${name}_p$i -- python $R/tools/prof_match.py 2 > $O/log_${name}_$i.txt 2>&1
  done
done
python - <<PY
import csv,glob,collections
for name in ("pipe","old","skel"):
    agg=collections.defaultdict(list); dur=[]
    for f in sorted(glob.glob("$O/%s_p*_counter_collection.csv"%name)):
        for r in csv.DictReader(open(f)):
            if "match_coarse" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob("$O/%s_p*_kernel_trace.csv"%name)):
        for r in csv.DictReader(open(f)):
            if "match_coarse" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    a={k:sum(v)/len(v) for k,v in agg.items()}
    d=sorted(dur)[len(dur)//2]
    cyc=a["GRBM_GUI_ACTIVE"]/8
    print(name, "dur_us %.0f"%d, "clock_GHz %.2f"%(cyc/d/1e3), "mfma_busy %.3f"%(a["SQ_VALU_MFMA_BUSY_CYCLES"]/(cyc*1024)),
          "wait_any %.2f wait_inst %.2f active %.2f wait_lds %.3f"%tuple(a[k]/a["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS")),
          "salu/mfma %.2f valu/mfma %.2f"%(a["SQ_INSTS_SALU"]/a["SQ_INSTS_MFMA"], a["SQ_INSTS_VALU"]/a["SQ_INSTS_MFMA"]))
PY
