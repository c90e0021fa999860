#!/usr/bin/env python
"""Copy the round-6 evidence of `bash tools/r06_final.sh` (gpurun_out/r06final/) into profiles/r06_* and write
profiles/r06_bench_summary.md from it.      python tools/refresh_profiles_r06.py [source directory]"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "r06final"
DST = ROOT / "profiles"


def last_json(p):
    return json.loads(Path(p).read_text().strip().splitlines()[-1])


def stats_table(path, n=16):
    if not Path(path).exists():
        return "(not collected)"
    rows = list(csv.DictReader(open(path)))
    lib = [r for r in rows if "anonymous namespace" in r["Name"] or "_GLOBAL__N_" in r["Name"]]
    out = ["| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in lib[:n]:
        name = r["Name"].replace("vfmm::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out)


def text(name, tail=None):
    p = DST / name
    if not p.exists():
        return "(not collected)"
    t = "\n".join(ln for ln in p.read_text().rstrip().splitlines() if "amdgpu.ids" not in ln)
    return t[-tail:] if tail else t


def stage_rows(st):
    if not st or "error" in st:
        return "(not collected)"
    out = ["| stage | ms alone | bound | achieved | fraction of the peak |", "|---|---|---|---|---|"]
    for k, v in st.items():
        if not isinstance(v, dict):
            continue
        ach = (f"{v['achieved_TBs']:.2f} TB/s" if "achieved_TBs" in v else f"{v['achieved_TFLOPs']:.0f} T(FL)OP/s" if "achieved_TFLOPs" in v else "-")
        fr = "-" if v.get("frac") is None else f"{v['frac']:.3f}"
        out.append(f"| {k} | {v['ms']:.3f} | {v['bound'].split(' (')[0]} | {ach} | {fr} |")
    out.append(f"| sum of the C2 stages | {st.get('sum_of_stages_ms', float('nan')):.3f} | | | |")
    return "\n".join(out)


def pmc_line(p):
    if not p:
        return "(not collected)"
    return (f"**{p['hbm_bytes_per_launch'] / 1e9:.3f} GB per launch** (FETCH_SIZE {p['FETCH_SIZE_KB'] / 1024:.0f} MB x 2 per the guide's gfx950 "
            f"correction + WRITE_SIZE {p['WRITE_SIZE_KB'] / 1024:.0f} MB), L2 hit rate {p['TCC_hit_rate']:.3f}, clock {p['clock_GHz']:.2f} GHz, "
            f"MFMA pipe busy {p['mfma_busy_fraction']:.3f} of all SIMD cycles, LDS array busy {p['lds_array_busy_fraction']:.3f}, per MFMA "
            f"{p['per_mfma']['valu_incl_mfma']:.2f} VALU (incl. the MFMA) / {p['per_mfma']['salu']:.2f} SALU / {p['per_mfma']['lds']:.2f} LDS; wave time "
            f"{p['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f} issuing / {p['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f} waiting to issue / "
            f"{p['wave_time_shares']['SQ_WAIT_ANY']:.2f} in waitcnt + barrier; median duration under the counters {p['median_duration_us_under_pmc']:.0f} us")


def main():
    copies = {"bench.json": "r06_bench.json", "bench_streams1.json": "r06_bench_streams1.json",
              "prof/bench_kernel_stats.csv": "r06_bench_kernel_stats.csv", "prof1/bench1_kernel_stats.csv": "r06_bench_streams1_kernel_stats.csv",
              "pytest_gpu.txt": "r06_pytest_gpu.txt", "smoke.txt": "r06_smoke.txt",
              "pmc_match_coarse_mx6.json": "r06_pmc_match_coarse_mx6.json", "pmc_match_coarse_mx6half.json": "r06_pmc_match_coarse_mx6half.json",
              "dev_mx6.txt": "r06_dev_mx6.txt", "ab_r4.txt": "r06_ab_records.txt", "soak_mx6.txt": "r06_soak_mx6.txt", "soak_half.txt": "r06_soak_half.txt",
              "pipeline_cycle.txt": "r06_pipeline_cycle.txt", "time_vit_batch.txt": "r06_time_vit_batch.txt", "prof_vit_batch.txt": "r06_prof_vit_batch.txt",
              "time_api.txt": "r06_time_api.txt", "time_c3_pipe.txt": "r06_time_c3_pipe.txt", "other_rows.txt": "r06_other_rows.txt",
              "time_pairs.txt": "r06_time_pairs.txt", "time_c3_modes.txt": "r06_time_c3_modes.txt", "neardup.json": "r06_neardup.json", "sweep_slices.txt": "r06_sweep_slices.txt", "trace_c3_pipe.txt": "r06_trace_c3_pipe.txt",
              "time_c3_group.txt": "r06_time_c3_group.txt", "ab_vit_astat.txt": "r06_ab_vit_astat.txt", "ab_prep_forms.txt": "r06_ab_prep_forms.txt", "vit_split.txt": "r06_vit_split.txt", "hbm_probe.txt": "r06_hbm_probe.txt", "prof_finish.txt": "r06_prof_finish.txt"}
    copies.update({"pmc_match_coarse_mx6fused.json": "r06_pmc_match_coarse_mx6fused.json", "prof_vit.txt": "r06_prof_vit.txt",
                   "pmc_vit_96images.json": "r06_pmc_vit_96images.json", "pmc_vit_90images.json": "r06_pmc_vit_90images.json",
                   "pmc_vit_6images.json": "r06_pmc_vit_6images.json", "mx6_probe.txt": "r06_mx6_probe.txt"})
    copies.update({"pmc_prep.txt": "r06_pmc_prep.txt", "dev_prep_once.txt": "r06_dev_prep_once.txt", "sweep_slices_alone.txt": "r06_sweep_slices_alone.txt",
                   "time_api_cold.txt": "r06_time_api_cold.txt", "trace_pipe_d2.txt": "r06_trace_pipe_d2.txt"})
    copies.update({"time_api_steps.txt": "r06_time_api_steps.txt", "ab_voxel_grid.txt": "r06_ab_voxel_grid.txt", "trace_voxel_grid.txt": "r06_trace_voxel_grid.txt",
                   "ab_api_search.txt": "r06_ab_api_search.txt", "ab_vit_wide.txt": "r06_ab_vit_wide.txt", "trace_vit_lds.txt": "r06_trace_vit_lds.txt",
                   "ab_vit_hot_a.txt": "r06_ab_vit_hot_a.txt", "f16_mfma_probe.txt": "r06_f16_mfma_probe.txt", "mfma_lds_probe.txt": "r06_mfma_lds_probe.txt", "l2_lds_probe.txt": "r06_l2_lds_probe.txt"})
    copies.update({"ab_vit_fused_qkv_sweep.txt": "r06_ab_vit_fused_qkv_sweep.txt", "trace_vit_fused.txt": "r06_trace_vit_fused.txt",
                   "prof_vit_84images.txt": "r06_prof_vit_84images.txt", "soak_vit_fused.txt": "r06_soak_vit_fused.txt", "pmc_vit_84images.json": "r06_pmc_vit_84images.json"})
    for i in range(1, 7):
        copies[f"pmc_vit6_pass{i}_counter_collection.csv"] = f"r06_pmc_vit6_pass{i}_counter_collection.csv"
    for i in range(1, 8):
        copies[f"pmc_mx6fused_pass{i}_counter_collection.csv"] = f"r06_pmc_mx6fused_pass{i}_counter_collection.csv"
        copies[f"pmc_mx6_pass{i}_counter_collection.csv"] = f"r06_pmc_mx6_pass{i}_counter_collection.csv"
        copies[f"pmc_mx6half_pass{i}_counter_collection.csv"] = f"r06_pmc_mx6half_pass{i}_counter_collection.csv"
    for a, b in copies.items():
        src = SRC / a
        if not src.exists() and "/" in a:   # rocprofv3 nests its output under the host name
            found = list((SRC / a.split("/")[0]).rglob(a.split("/")[1]))
            src = found[0] if found else src
        if src.exists():
            shutil.copy(src, DST / b)
    # evidence gathered earlier in the round by separate calls (kept under their own names)
    b = last_json(DST / "r06_bench.json")
    b1 = last_json(DST / "r06_bench_streams1.json") if (DST / "r06_bench_streams1.json").exists() else None
    pm6 = json.loads((DST / "r06_pmc_match_coarse_mx6.json").read_text()) if (DST / "r06_pmc_match_coarse_mx6.json").exists() else None
    pm6h = json.loads((DST / "r06_pmc_match_coarse_mx6half.json").read_text()) if (DST / "r06_pmc_match_coarse_mx6half.json").exists() else None
    pm6f = json.loads((DST / "r06_pmc_match_coarse_mx6fused.json").read_text()) if (DST / "r06_pmc_match_coarse_mx6fused.json").exists() else None

    def vit_pmc(name):
        p = DST / name
        if not p.exists():
            return "(not collected)"
        a = json.loads(p.read_text())
        out = ["| kernel | median us | MFMA busy | per wave: MFMA | VALU | SALU | VALU / MFMA | SALU / MFMA | issuing / waiting to issue / waitcnt | fetched MB (x2) | written MB |", "|---|---|---|---|---|---|---|---|---|---|---|"]
        for k, z in a.items():
            i = z["insts_per_wave"]
            mf = max(i["SQ_INSTS_MFMA"], 1e-9)
            sh = z["wave_time_shares"]
            out.append(f"| {k} | {z['median_us']:.1f} | {z['mfma_busy_fraction_of_all_simd_cycles']:.3f} | {i['SQ_INSTS_MFMA']:.0f} | {i['SQ_INSTS_VALU']:.0f} | {i['SQ_INSTS_SALU']:.0f} | "
                       + (f"{(i['SQ_INSTS_VALU'] - i['SQ_INSTS_MFMA']) / mf:.1f} | {i['SQ_INSTS_SALU'] / mf:.1f}" if i["SQ_INSTS_MFMA"] > 0 else "- | -")
                       + f" | {sh['SQ_ACTIVE_INST_ANY']:.2f} / {sh['SQ_WAIT_INST_ANY']:.2f} / {sh['SQ_WAIT_ANY']:.2f} | {z['fetch_MB_x2']:.0f} | {z['write_MB']:.0f} |")
        return "\n".join(out)
    r = b["roofline"]
    ex = b.get("extra", {})
    cfg = b["config"]

    def variant(k):
        v = ex.get(k)
        if not v:
            return f"`extra.{k}`: (absent)"
        rl = v.get("roofline") or {}
        which = "fp6" if rl.get("peak", 0) > 6000 else "int8"
        return (f"`extra.{k}`: **{v.get('value', float('nan')):.1f} registrations/s** ({v.get('ms_per_step', float('nan')):.3f} ms)"
                + (f", coarse kernel {rl.get('avg_launch_ms', float('nan')):.3f} ms = {rl.get('frac', float('nan')):.3f} of the {which} peak" if rl else "")
                + (f", pass in use: {v.get('coarse_pass')}" if v.get("coarse_pass") else ""))

    c3, c3p, vb, api, c5, a6 = (ex.get(k, {}) for k in ("C3", "C3_pipelined", "ViT_batched", "API_ransac_registration", "C5", "A6_mutual_l2"))
    nd_txt = "(not collected)"
    if (DST / "r06_neardup.json").exists():
        nd = json.loads((DST / "r06_neardup.json").read_text())
        maps = []
        for k in nd:
            name = k.split(" | ")[0]
            if name not in maps:
                maps.append(name)
        modes = [c for c in ("mx6-half", "int8-half", "int8", "mx6", "int8-top2", "fp16") if all(f"{m} | {c}" in nd for m in maps)]
        nd_txt = ("| map | auto | " + " | ".join(modes) + " | same correspondences + pose | all-pairs fallbacks |\n|---|---|" + "---|" * len(modes) + "---|---|\n"
                  + "\n".join(
                      f"| {name} | {nd[name + ' | auto']['ms_per_registration']:.2f} ({nd[name + ' | auto']['pass_in_use']}"
                      f"{'' if nd[name + ' | auto']['pass_in_use'] == 'fp16' else ', ' + str(nd[name + ' | auto'].get('records_in_use', '?'))}) | "
                      + " | ".join(f"{nd[name + ' | ' + c]['ms_per_registration']:.2f}" for c in modes) + " | "
                      f"{all(nd[name + ' | ' + c]['same_result_as_auto'] for c in modes)} | "
                      f"{sum(nd[name + ' | ' + c]['fallback_queries'] for c in ['auto'] + modes)} |" for name in maps))
    md = f"""# Round 6 -- measurements on one MI355X (config C2: 20 000 x 200 000 x 384, 50 000 RANSAC iterations)

Produced by `bash tools/r06_final.sh` through `gpurun` (a fresh box per call; boxes of the pool differ by up to ~15 %),
collected by `python tools/refresh_profiles_r06.py`.  Raw files are next to this one (`r06_*`).  GPU suite on the same box:
`{text('r06_pytest_gpu.txt').splitlines()[-1]}`; `__graft_entry__.smoke()`: `{text('r06_smoke.txt').splitlines()[-1]}`.

## bench.py (default: `auto` -- on D.2 descriptors the half-width pass in fp6 with the bound test inside the kernel, record kind 8)

`python bench.py` -> `profiles/r06_bench.json`: **{b['value']:.1f} registrations/s** ({b['ms_per_step']:.3f} ms per
registration), dominant kernel `{r['kernel'].split(' (')[0]}` {r['avg_launch_ms']:.3f} ms per launch inside the timed region =
{r['achieved']:.0f} TOP/s = {r['frac']:.3f} of {r['peak'] / 1000:.1f} POP/s ({r.get('peak_note', 'dense MFMA peak')}; operations of the kernel as launched:
{r['flops_per_launch'] / 1e12:.3f} TOP -- coarse pass in use: {cfg.get('coarse_pass', '?')}; surviving chunks per query of the half-width
selection: {cfg.get('half_width_survivors_per_query')}); alone on the GPU {r['single_stream']['avg_launch_ms']:.3f} ms =
{r['single_stream']['achieved']:.0f} TOP/s = {r['single_stream']['frac']:.3f}.  `roofline.traffic` = {r.get('traffic')} bytes per launch
({r.get('traffic_source', 'PMC passes below')}).  Peak HBM allocated: {cfg.get('hbm_peak_allocated_gb', float('nan')):.1f} GB.
CPU oracle on the same box ({b['cpu_baseline']['cores']} threads): {b['cpu_baseline']['value']:.3f} registrations/s.
Pose delta vs the oracle on identical inputs (`extra.pose_delta_vs_oracle`): {ex.get('pose_delta_vs_oracle', {}).get('pose_delta_vs_oracle_frobenius')}.

The same pipeline with the coarse pass pinned, other data, other rows (same process, same box):

- {variant('C2_sustained')} -- {ex.get('C2_sustained', {}).get('steps', '?')} steps instead of 20
- {variant('C2_half_width_mx6')}
- {variant('C2_half_width_int8')}
- {variant('C2_full_width_mx6')} -- every column in the coarse pass, nothing depends on how the descriptors prune
- {variant('C2_full_width_mx6_fused')} -- the same product with the gate test in the kernel's epilogue (record kind 10: no record array, no selection sweep)
- {variant('C2_full_width')}
- {variant('C2_lifted')} -- map descriptors lifted from overlapping patch grids (near-duplicates), policy by feedback
- `extra.A6_mutual_l2`: {a6.get('ms_mutual_pairs', float('nan')):.2f} ms per `find_correspondences(mutual_filter=True)` at C2 size ({a6.get('mutual_pairs')} mutual pairs)
- `extra.C3`: {c3.get('ms_end_to_end', float('nan')):.2f} ms end to end, one pair at a time (ViT {c3.get('ms_vit', float('nan')):.3f}, project + lift {c3.get('ms_project_lift', float('nan')):.3f}, registration {c3.get('ms_registration', float('nan')):.2f}; ViT at {c3.get('vit_roofline', {}).get('frac', float('nan')):.3f} of the fp16 MFMA peak)
- `extra.C3_pipelined`: **{c3p.get('value', float('nan')):.1f} registrations/s** from uint8 images ({c3p.get('ms_per_step', float('nan')):.3f} ms per pair; feature stage of pair i + 1 beside the registration of pair i)
- `extra.C3_pipelined.grouped`: **{c3p.get('grouped', {}).get('value', float('nan')):.1f} registrations/s** with the cameras of {c3p.get('grouped', {}).get('pairs_per_vit_call')} pairs per ViT call (`EndToEndPipeline.submit_group`; {c3p.get('grouped', {}).get('ms_per_step', float('nan')):.3f} ms per pair); `tools/time_c3_group.py` (`r06_time_c3_group.txt`, pairs per call against registrations/s, a process of its own):

```
{text('r06_time_c3_group.txt')}
```

- `extra.ViT_batched`: {vb.get('images')} images per call {vb.get('ms', float('nan')):.2f} ms = {vb.get('ms_per_scan_of_6', float('nan')):.3f} ms per scan of 6 ({vb.get('roofline', {}).get('achieved', float('nan')):.0f} TFLOP/s); 84 images {vb.get('at_84_images', {}).get('ms', float('nan')):.2f} ms, 90 images {vb.get('at_90_images', {}).get('ms', float('nan')):.2f} ms, 48 images {vb.get('at_48_images', {}).get('ms', float('nan')):.2f} ms
- `extra.API_ransac_registration`: {api.get('ms_without_icp', float('nan')):.2f} ms numpy in / numpy out, {api.get('ms_with_icp', float('nan')):.2f} ms with the ICP refinement (scene's map kept between scans)
- `extra.C5.fp16_descriptor_storage` (the map held in fp16, rows widened on load): {json.dumps(c5.get('fp16_descriptor_storage'))}
- `extra.stages` RANSAC on executed operations: {json.dumps(ex.get('stages', {}).get('RANSAC + Kabsch (50 000 hypotheses, fp64)', {}).get('executed'))}
- `extra.C5` (50k x 1M x 768; pass in use: {c5.get('coarse_pass', '?')}): coarse kernel {c5.get('ms_coarse_kernel', float('nan')):.2f} ms = {c5.get('roofline', {}).get('frac', float('nan')):.3f} of {c5.get('roofline', {}).get('peak', 0) / 1000:.0f} P(FL)OP/s, registration {c5.get('ms_registration', float('nan')):.2f} ms

Every stage alone on the GPU with SURVEY 8 D.3's algorithmic work and the peak that bounds it (`extra.stages`; in the pipeline the stages overlap):

{stage_rows(ex.get('stages', {}))}

`python bench.py --streams 1` (every kernel serialised on one stream) -> `profiles/r06_bench_streams1.json`:
{(f"{b1['value']:.1f} registrations/s, dominant kernel {b1['roofline']['avg_launch_ms']:.3f} ms") if b1 else '(not collected)'}.

## rocprofv3 --kernel-trace --stats of the default bench command

`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --no-cpu-baseline --no-extra`
-> `profiles/r06_bench_kernel_stats.csv` (library kernels only; 20 timed + 3 warm-up registrations + the isolated launches of
`single_stream`; the solve stages overlap the coarse pass, so a solve kernel's duration includes waiting for compute units
held by the coarse kernel):

{stats_table(DST / 'r06_bench_kernel_stats.csv', 22)}

Serial (`--streams 1`), `profiles/r06_bench_streams1_kernel_stats.csv`:

{stats_table(DST / 'r06_bench_streams1_kernel_stats.csv', 20)}

## PMC passes of the coarse kernel (`bash tools/pmc_coarse.sh`, separate --pmc passes, --kernel-trace only)

fp6 half-width kernel with the bound test inside (`VFM_RECORDS=8`: what the default bench runs on D.2 data;
`profiles/r06_pmc_match_coarse_mx6half.json` + `profiles/r06_pmc_mx6half_pass*_counter_collection.csv`,
{pm6h['kernel'] if pm6h else '?'}): {pmc_line(pm6h)}.

fp6 full-width kernel with the gate test in its epilogue (`VFM_RECORDS=10`, round 6; `profiles/r06_pmc_match_coarse_mx6fused.json`,
{pm6f['kernel'] if pm6f else '?'}): {pmc_line(pm6f)}.

fp6 full-width kernel (`VFM_RECORDS=5`; `profiles/r06_pmc_match_coarse_mx6.json` + `profiles/r06_pmc_mx6_pass*_counter_collection.csv`,
{pm6['kernel'] if pm6 else '?'}): {pmc_line(pm6)}.

## Operand preparation: one read of the rows (round 6; `tools/pmc_prep.sh`, `tools/dev_prep_once.py`, `tools/ab_prep_r6.py`)

HBM traffic per launch of the three forms (separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes, gfx950 correction applied):

```
{text('r06_pmc_prep.txt', 2500)}
```

Checks and times alone (`r06_dev_prep_once.txt`, tail) and inside the pipeline (`r06_ab_prep_forms.txt`):

```
{text('r06_dev_prep_once.txt', 1400)}
{text('r06_ab_prep_forms.txt', 3000)}
```

The headline's coarse kernel alone against the number of map slices (`tools/sweep_slices_r6.py`; the launcher's rule = slices 0):

```
{text('r06_sweep_slices_alone.txt', 1500)}
```

Kernel timeline of the pipeline (`tools/trace_pipe.sh mx6-half d2`, under the profiler):

```
{text('r06_trace_pipe_d2.txt', 4000)}
```

The reference-shaped call cold (default node: the map rebuilt per call), `set_map()`, and through the handle (`tools/time_api_cold.py`):

```
{text('r06_time_api_cold.txt', 2000)}
```

## fp6 MFMA shapes, bare (`tools/probe/mx6_probe.hip`, VERDICT r4 item 3)

```
{text('r06_mx6_probe.txt')}
```

## ViT-S/14: one scan, batches, kernel by kernel (`tools/time_vit_batch.py`, `tools/prof_vit_r06.sh`, `tools/pmc_vit.sh`)

```
{text('r06_time_vit_batch.txt', 2500)}
```

```
{text('r06_prof_vit.txt', 6000)}
```

Counters per kernel (separate `--pmc` passes, `--kernel-trace` only), 96 images per call (`r06_pmc_vit_96images.json`; LDS-tiled GEMMs):

{vit_pmc('r06_pmc_vit_96images.json')}

90 images per call (`r06_pmc_vit_90images.json`; QKV / fc1 by the token-stationary kernel):

{vit_pmc('r06_pmc_vit_90images.json')}

6 images per call (`r06_pmc_vit_6images.json` + `r06_pmc_vit6_pass*_counter_collection.csv`):

{vit_pmc('r06_pmc_vit_6images.json')}

## The reference-shaped API (`tools/time_api.py`)

```
{text('r06_time_api.txt', 2500)}
```

Step by step (`tools/time_api_steps.py`), and VoxelDownsample alone: the general multi-launch path against the one-launch kernel (`tools/ab_voxel_grid.py`),
the kernel phase by phase (`tools/trace_voxel_grid.py`), the search of ~10^3 queries at half / full width (`tools/ab_api_search.py`):

```
{text('r06_time_api_steps.txt', 2500)}
{text('r06_ab_voxel_grid.txt', 2500)}
{text('r06_trace_voxel_grid.txt', 2500)}
{text('r06_ab_api_search.txt', 2500)}
```

(The first one-launch attempt of the round -- one workgroup going on alone -- measured slower and is gone: `r06_time_api_onelaunch.txt`.)

## What bounds the ViT GEMMs (`tools/probe/f16_mfma_probe.hip`, `tools/probe/l2_lds_probe.hip`, `tools/trace_vit_lds.py`, `tools/ab_vit_hot_a.sh`, `tools/ab_vit_wide.py`)

```
{text('r06_f16_mfma_probe.txt', 2500)}
{text('r06_mfma_lds_probe.txt', 4000)}
{text('r06_l2_lds_probe.txt', 4000)}
{text('r06_trace_vit_lds.txt', 2500)}
{text('r06_ab_vit_hot_a.txt', 2500)}
{text('r06_ab_vit_wide.txt', 2500)}
```

The finish stage kernel by kernel (`tools/prof_finish.sh`; lifted + common descriptors behind the fp6 / int8 full-width pass, D.2 behind the fused half-width pass):

```
{text('r06_prof_finish.txt', 2500)}
```

## F rows, C3 stages, RANSAC alone, row A6 (`r06_other_rows.txt`)

```
{text('r06_other_rows.txt')}
```

## Soaks beyond the suite's fixed seeds

`python tools/soak_mx6.py 40 505` (fp6 kinds 5, 6, 7, 8, 9, 10 against best-score int8 records): `{text('r06_soak_mx6.txt').splitlines()[-1]}`;
`python tools/soak_half.py 40 505`: `{text('r06_soak_half.txt').splitlines()[-1]}`.
"""
    (DST / "r06_bench_summary.md").write_text(md)
    print(md[:2500])


if __name__ == "__main__":
    main()
