#!/usr/bin/env python
"""Copy the round-2 evidence of `bash tools/r02_final.sh` (gpurun_out/r02final/) into profiles/r02_* and write
profiles/r02_bench_summary.md from it."""
import csv
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "r02final"
DST = ROOT / "profiles"


def last_json(p):
    return json.loads(Path(p).read_text().strip().splitlines()[-1])


def stats_table(path, n=16):
    rows = list(csv.DictReader(open(path)))
    lib = [r for r in rows if "anonymous namespace" in r["Name"] or "_GLOBAL__N_" in r["Name"]]
    out = ["| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in lib[:n]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out)


def main():
    copies = {"bench.json": "r02_bench.json", "bench_streams1.json": "r02_bench_streams1.json",
              "bench_f16_same_box.json": "r02_bench_f16_same_box.json",
              "prof/bench_kernel_stats.csv": "r02_bench_kernel_stats.csv",
              "prof1/bench1_kernel_stats.csv": "r02_bench_streams1_kernel_stats.csv",
              "pmc_match_coarse.json": "r02_pmc_match_coarse_i8.json", "pytest_gpu.txt": "r02_pytest_gpu.txt",
              "neardup.json": "r02_neardup.json", "time_prep.txt": "r02_time_prep.txt",
              "steps_sweep.txt": "r02_steps_sweep.txt", "power_probe.txt": "r02_power_probe.txt",
              "time_vit_batch.txt": "r02_time_vit_batch.txt", "time_ungated.txt": "r02_time_ungated.txt",
              "time_c3_modes.txt": "r02_time_c3_modes.txt", "tax_probe.txt": "r02_tax_probe.txt",
              "cu_mask_probe.txt": "r02_cu_mask_probe.txt", "prof_search_c2.txt": "r02_prof_search_c2.txt",
              "prof_c3.txt": "r02_prof_c3.txt", "ab_half.txt": "r02_ab_half.txt",
              "pmc_match_coarse_half.json": "r02_pmc_match_coarse_i8half.json",
              "bench_int8_full_same_box.json": "r02_bench_int8_full_same_box.json"}
    for i in range(1, 8):
        copies[f"pmc_pass{i}_counter_collection.csv"] = f"r02_pmc_pass{i}_counter_collection.csv"
        copies[f"pmc_half_pass{i}_counter_collection.csv"] = f"r02_pmc_half_pass{i}_counter_collection.csv"
    for a, b in copies.items():
        if (SRC / a).exists():
            shutil.copy(SRC / a, DST / b)
    b = last_json(DST / "r02_bench.json")
    b1 = last_json(DST / "r02_bench_streams1.json")
    bf = last_json(DST / "r02_bench_f16_same_box.json")
    bi = last_json(DST / "r02_bench_int8_full_same_box.json")
    pmh = json.loads((DST / "r02_pmc_match_coarse_i8half.json").read_text())
    pmc = json.loads((DST / "r02_pmc_match_coarse_i8.json").read_text())
    r = b["roofline"]
    ex = b["extra"]
    nd = json.loads((DST / "r02_neardup.json").read_text())
    maps = []
    for k in nd:
        name = k.split(" | ")[0]
        if name not in maps:
            maps.append(name)
    nd_rows = "\n".join(
        f"| {name} | {nd[name + ' | auto']['ms_per_registration']:.2f} ({nd[name + ' | auto']['pass_in_use']}"
        f"{'' if nd[name + ' | auto']['pass_in_use'] == 'fp16' else ', ' + nd[name + ' | auto'].get('records_in_use', '?')}) | "
        f"{nd[name + ' | int8-half']['ms_per_registration']:.2f} | "
        f"{nd[name + ' | int8']['ms_per_registration']:.2f} | {nd[name + ' | int8-top2']['ms_per_registration']:.2f} | "
        f"{nd[name + ' | fp16']['ms_per_registration']:.2f} | "
        f"{all(nd[name + ' | ' + c]['same_result_as_auto'] for c in ('int8-half', 'int8', 'int8-top2', 'fp16'))} | "
        f"{sum(nd[name + ' | ' + c]['fallback_queries'] for c in ('auto', 'int8-half', 'int8', 'int8-top2', 'fp16'))} |" for name in maps)
    md = f"""# Round 2 -- measurements on one MI355X (config C2: 20 000 x 200 000 x 384, 50 000 RANSAC iterations)

Produced by `bash tools/r02_final.sh` (GPU tests, smoke, bench default / fp16 pass / serial / under rocprofv3, PMC passes of
the coarse kernel, duplicate-rich maps) through `gpurun` (a fresh box per call; boxes of the pool differ by a few per cent),
collected by `python tools/refresh_profiles_r02.py`.  Raw files are next to this one (`r02_*`); the same set for the fp16
coarse pass as it stood before the int8 pass is in `r02_*_f16*` (summary: `r02_bench_summary_f16.md`).

## bench.py (default: int8 coarse pass -- on D.2 descriptors the half-width pass; operand preparation | coarse pass | two solve streams)

`python bench.py` -> `profiles/r02_bench.json`: **{b['value']:.1f} registrations/s** ({b['ms_per_step']:.3f} ms per
registration), dominant kernel `{r['kernel'].split(' (')[0]}` {r['avg_launch_ms']:.3f} ms per launch inside the timed region =
{r['achieved']:.0f} TOP/s = {r['frac']:.3f} of the {r['peak'] / 1000:.1f} POP/s dense int8 MFMA peak (operations of the kernel as launched:
{r['flops_per_launch'] / 1e12:.3f} TOP -- coarse pass in use: {b['config'].get('coarse_pass', '?')});
alone on the GPU {r['single_stream']['avg_launch_ms']:.3f} ms = {r['single_stream']['achieved']:.0f} TOP/s =
{r['single_stream']['frac']:.3f}.  CPU oracle on the same box ({b['cpu_baseline']['cores']} threads): {b['cpu_baseline']['value']:.3f} registrations/s.
Pose delta vs the oracle on identical inputs (`extra.pose_delta_vs_oracle`): {ex.get('pose_delta_vs_oracle', {}).get('pose_delta_vs_oracle_frobenius')}.
`extra.C3`: {ex['C3']['ms_end_to_end']:.2f} ms end to end (ViT {ex['C3']['ms_vit']:.2f}, project + lift {ex['C3']['ms_project_lift']:.2f}, registration {ex['C3']['ms_registration']:.2f}; ViT at {ex['C3']['vit_roofline']['frac']:.3f} of the MFMA peak).
`extra.C5` (50k x 1M x 768, int8 pass): coarse kernel {ex['C5']['ms_coarse_kernel']:.1f} ms = {ex['C5']['roofline']['frac']:.3f} of the int8 peak, registration {ex['C5']['ms_registration']:.1f} ms.

Same box, `VFM_COARSE=int8 python bench.py` (full-width int8 pass, best-score records) -> `profiles/r02_bench_int8_full_same_box.json`:
{bi['value']:.1f} registrations/s, kernel {bi['roofline']['avg_launch_ms']:.3f} ms ({bi['roofline']['frac']:.3f} of the int8 peak).

Same box, `VFM_VARIANT=5 python bench.py` (the fp16 coarse pass with sparse records, round 2's earlier default) ->
`profiles/r02_bench_f16_same_box.json`: {bf['value']:.1f} registrations/s, kernel {bf['roofline']['avg_launch_ms']:.3f} ms
({bf['roofline']['frac']:.3f} of the fp16 peak).

`python bench.py --streams 1` (every kernel serialised on one stream) -> `profiles/r02_bench_streams1.json`:
{b1['value']:.1f} registrations/s, dominant kernel {b1['roofline']['avg_launch_ms']:.3f} ms.

## rocprofv3 --kernel-trace --stats of the default bench command

`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --no-cpu-baseline --no-extra`
-> `profiles/r02_bench_kernel_stats.csv` (library kernels only; 20 timed + 3 warm-up registrations + the 6 isolated
launches of `single_stream`; the solve stages overlap the coarse pass, so a solve kernel's duration includes waiting for
compute units held by the coarse kernel):

{stats_table(DST / 'r02_bench_kernel_stats.csv', 18)}

Serial (`--streams 1`), `profiles/r02_bench_streams1_kernel_stats.csv`:

{stats_table(DST / 'r02_bench_streams1_kernel_stats.csv', 16)}

## PMC passes of the coarse kernel (`bash tools/pmc_coarse.sh`, separate --pmc passes, --kernel-trace only)

Half-width kernel (`VFM_RECORDS=3`; `profiles/r02_pmc_match_coarse_i8half.json` + `profiles/r02_pmc_half_pass*_counter_collection.csv`,
{pmh['kernel']}): **{pmh['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch** (`roofline.traffic` of the default bench line), L2 hit rate
{pmh['TCC_hit_rate']:.3f}, clock {pmh['clock_GHz']:.2f} GHz, MFMA pipe busy {pmh['mfma_busy_fraction']:.3f}, LDS array busy {pmh['lds_array_busy_fraction']:.3f},
per MFMA {pmh['per_mfma']['valu_incl_mfma']:.2f} VALU (incl. the MFMA) / {pmh['per_mfma']['salu']:.2f} SALU / {pmh['per_mfma']['lds']:.2f} LDS.

Full-width kernel (`VFM_RECORDS=0`):

`profiles/r02_pmc_match_coarse_i8.json` + `profiles/r02_pmc_pass*_counter_collection.csv` ({pmc['kernel']}): FETCH_SIZE {pmc['FETCH_SIZE_KB'] / 1024:.0f} MB (x2 per the
guide's gfx950 correction), WRITE_SIZE {pmc['WRITE_SIZE_KB'] / 1024:.0f} MB -> **{pmc['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch** (`roofline.traffic`; fp16 pass: 1.05 GB);
L2 hit rate {pmc['TCC_hit_rate']:.3f}; clock {pmc['clock_GHz']:.2f} GHz; MFMA pipe busy {pmc['mfma_busy_fraction']:.3f} of all SIMD cycles; LDS array busy
{pmc['lds_array_busy_fraction']:.3f}; per MFMA {pmc['per_mfma']['valu_incl_mfma']:.2f} VALU (incl. the MFMA), {pmc['per_mfma']['salu']:.2f} SALU, {pmc['per_mfma']['lds']:.2f} LDS; wave time
{pmc['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f} issuing / {pmc['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f} waiting to issue / {pmc['wave_time_shares']['SQ_WAIT_ANY']:.2f} in waitcnt + barrier.

## Duplicate-rich maps (`python tools/time_neardup.py`, C2 size, the bench's pipeline; `r02_neardup.json`)

ms per registration with the coarse pass chosen by the pipeline's feedback (`auto`: the pass and record kind in use after the
warm-up in brackets), and with each mode forced (int8-half = the half-width pass, int8 = best-score records, int8-top2 = packed
top-2 records):

| map | auto | int8-half | int8 | int8-top2 | fp16 | same correspondences + pose | all-pairs fallbacks |
|---|---|---|---|---|---|---|---|
{nd_rows}

## Second half of the round (files next to this one)

`r02_steps_sweep.txt` (registrations/s against the number of timed steps, with the coarse kernel's duration per step: the
first ~15 launches after the synchronise run slower), `r02_power_probe.txt` (rocm-smi power / clocks under an 8000-step
run), `r02_tax_probe.txt` (steady state with side stages replaced by no-ops), `r02_cu_mask_probe.txt` (side stages on
CU-masked streams: slower in every split), `r02_prof_search_c2.txt` (kernels of one gated C2 search), `r02_prof_c3.txt` +
`r02_time_c3_modes.txt` (C3's registration per coarse mode), `r02_time_ungated.txt` (ungated calls: fp16 pass vs the
routing by size), `r02_time_vit_batch.txt` (ViT forward against the number of images per call), `r02_ab_half.txt` (half-width pass vs
best-score / top-2 records at four sizes: coarse kernel, finish stage, survivors, agreement of the answers).

## Other evidence files

`r02_admissible_c2.json` (every C2 row is an admissible fp32 IndexFlatIP answer), `r02_c3_vit_precision.json` (fp16 ViT vs
fp32 oracle ViT: keep set, arg-max flips, pose), `r02_pmc_vit.json` + `r02_vit_kernel_stats.csv` (ViT kernels: wave lifetime
vs kernel duration), `r02_time_prep.txt` (operand preparation alone), `r02_pytest_gpu.txt`.
"""
    (DST / "r02_bench_summary.md").write_text(md)
    print(md[:1800])


if __name__ == "__main__":
    main()
