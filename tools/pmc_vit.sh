# PMC passes for the ViT kernels (\$VIT_IMAGES x 1200 x 1600, default 6, ViT-S/14) -> gpurun_out/pmc_vit/summary.json
# Separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_vit
rm -rf $O && mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p$i -- python $R/tools/prof_vit.py ${DEFAULT_TILES:-1} 3 ${VIT_IMAGES:-6} > $O/log$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def short(n):
    names = {"vit_gemm_kernel<1": "gemm QKV", "vit_gemm_kernel<2": "gemm proj/fc2 (+residual)", "vit_gemm_kernel<3": "gemm fc1 (+GELU)", "vit_gemm_kernel<0": "gemm patch embed",
             "vit_gemm_lds_kernel<1": "gemm QKV (LDS-tiled)", "vit_gemm_lds_kernel<2": "gemm proj/fc2 (+residual) (LDS-tiled)", "vit_gemm_lds_kernel<3": "gemm fc1 (+GELU) (LDS-tiled)",
             "vit_gemm_lds_kernel<0": "gemm patch embed (LDS-tiled)", "vit_attention_lds": "attention (K / V^T in the LDS)", "vit_qkv_attention": "QKV + attention, one workgroup per (image, head)",
             "vit_gemm_astat2_kernel<3": "gemm fc1 (+GELU) (token-stationary, two tiles)", "vit_gemm_astat2_kernel<1": "gemm QKV (token-stationary, two tiles)"}
    for k in ("vit_qkv_attention", "vit_gemm_astat2_kernel<3", "vit_gemm_astat2_kernel<1", "vit_gemm_lds_kernel<1", "vit_gemm_lds_kernel<2", "vit_gemm_lds_kernel<3", "vit_gemm_lds_kernel<0", "vit_gemm_kernel<1", "vit_gemm_kernel<2", "vit_gemm_kernel<3", "vit_gemm_kernel<0", "vit_attention_lds", "vit_attention", "vit_layernorm", "vit_final", "vit_preprocess", "vit_gemm64", "vit_gemm_row"):
        if k in n: return names.get(k, k)
    return None
for f in sorted(glob.glob("$O/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$O/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, cs in agg.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    d = sorted(dur[k])[len(dur[k]) // 2]
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    out[k] = {"median_us": d, "clock_GHz": cyc / d / 1e3, "waves": a.get("SQ_WAVES"),
              "mfma_busy_fraction_of_all_simd_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
              "wave_time_shares": {c: a[c] / a["SQ_WAVE_CYCLES"] for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
              "mean_wave_lifetime_us": a["SQ_WAVE_CYCLES"] / max(a.get("SQ_WAVES", 1), 1) / (cyc / d) ,
              "insts_per_wave": {c: a[c] / max(a.get("SQ_WAVES", 1), 1) for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")},
              "fetch_MB_x2": 2 * a["FETCH_SIZE"] / 1024, "write_MB": a["WRITE_SIZE"] / 1024,
              "l2_hit_rate": a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1)}
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
