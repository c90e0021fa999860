#!/usr/bin/env python
"""Copy the round-3 evidence of `bash tools/r03_final.sh` (gpurun_out/r03final/) into profiles/r03_* and write
profiles/r03_bench_summary.md from it."""
import csv
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "r03final"
DST = ROOT / "profiles"


def last_json(p):
    return json.loads(Path(p).read_text().strip().splitlines()[-1])


def stats_table(path, n=16):
    rows = list(csv.DictReader(open(path)))
    lib = [r for r in rows if "anonymous namespace" in r["Name"] or "_GLOBAL__N_" in r["Name"]]
    out = ["| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in lib[:n]:
        name = r["Name"].replace("vfmm::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out)


def text(name):
    p = DST / name
    return p.read_text().rstrip() if p.exists() else "(not collected)"


def pmc_line(p):
    return (f"**{p['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch** (FETCH_SIZE {p['FETCH_SIZE_KB'] / 1024:.0f} MB x 2 per the guide's gfx950 "
            f"correction + WRITE_SIZE {p['WRITE_SIZE_KB'] / 1024:.0f} MB), L2 hit rate {p['TCC_hit_rate']:.3f}, clock {p['clock_GHz']:.2f} GHz, "
            f"MFMA pipe busy {p['mfma_busy_fraction']:.3f} of all SIMD cycles, LDS array busy {p['lds_array_busy_fraction']:.3f}, per MFMA "
            f"{p['per_mfma']['valu_incl_mfma']:.2f} VALU (incl. the MFMA) / {p['per_mfma']['salu']:.2f} SALU / {p['per_mfma']['lds']:.2f} LDS; wave time "
            f"{p['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f} issuing / {p['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f} waiting to issue / "
            f"{p['wave_time_shares']['SQ_WAIT_ANY']:.2f} in waitcnt + barrier; median duration under the counters {p['median_duration_us_under_pmc']:.0f} us")


def main():
    copies = {"bench.json": "r03_bench.json", "bench_streams1.json": "r03_bench_streams1.json",
              "prof/bench_kernel_stats.csv": "r03_bench_kernel_stats.csv",
              "prof1/bench1_kernel_stats.csv": "r03_bench_streams1_kernel_stats.csv",
              "pmc_match_coarse.json": "r03_pmc_match_coarse_i8.json", "pytest_gpu.txt": "r03_pytest_gpu.txt",
              "pmc_match_coarse_half.json": "r03_pmc_match_coarse_i8half.json",
              "bench_int8_full_same_box.json": "r03_bench_int8_full_same_box.json",
              "neardup.json": "r03_neardup.json", "time_pairs.txt": "r03_time_pairs.txt", "prof_pairs.txt": "r03_prof_pairs.txt",
              "other_rows.txt": "r03_other_rows.txt", "time_c3_modes.txt": "r03_time_c3_modes.txt",
              "ab_vit_xcd.txt": "r03_ab_vit_xcd.txt", "prof_c3_one.txt": "r03_prof_c3_one.txt",
              "pmc_match_coarse_mx6.json": "r03_pmc_match_coarse_mx6.json", "pmc_match_coarse_mx6half.json": "r03_pmc_match_coarse_mx6half.json",
              "queue_probe.txt": "r03_queue_probe.txt", "dev_mx6.txt": "r03_dev_mx6.txt",
              "ab_mx6_bench.txt": "r03_ab_mx6_bench.txt", "soak_mx6.txt": "r03_soak_mx6.txt", "pipeline_cycle.txt": "r03_pipeline_cycle.txt", "soak_half.txt": "r03_soak_half.txt", "soak_match.txt": "r03_soak_match.txt", "mx6_probe.txt": "r03_mx6_probe.txt",
              "lifted_stats.txt": "r03_lifted_stats.txt", "prof_finish.txt": "r03_prof_finish.txt", "lifted_cycle.txt": "r03_lifted_cycle.txt", "warmup_ab.txt": "r03_warmup_ab.txt"}
    for i in range(1, 8):
        copies[f"pmc_pass{i}_counter_collection.csv"] = f"r03_pmc_pass{i}_counter_collection.csv"
        copies[f"pmc_half_pass{i}_counter_collection.csv"] = f"r03_pmc_half_pass{i}_counter_collection.csv"
        copies[f"pmc_mx6_pass{i}_counter_collection.csv"] = f"r03_pmc_mx6_pass{i}_counter_collection.csv"
        copies[f"pmc_mx6half_pass{i}_counter_collection.csv"] = f"r03_pmc_mx6half_pass{i}_counter_collection.csv"
    for a, b in copies.items():
        src = SRC / a
        if not src.exists() and "/" in a:   # rocprofv3 nests its output under the host name
            found = list((SRC / a.split("/")[0]).rglob(a.split("/")[1]))
            src = found[0] if found else src
        if src.exists():
            shutil.copy(src, DST / b)
    b = last_json(DST / "r03_bench.json")
    b1 = last_json(DST / "r03_bench_streams1.json")
    bi = last_json(DST / "r03_bench_int8_full_same_box.json")
    pmh = json.loads((DST / "r03_pmc_match_coarse_i8half.json").read_text())
    pmc = json.loads((DST / "r03_pmc_match_coarse_i8.json").read_text())
    pm6 = json.loads((DST / "r03_pmc_match_coarse_mx6.json").read_text()) if (DST / "r03_pmc_match_coarse_mx6.json").exists() else None
    pm6h = json.loads((DST / "r03_pmc_match_coarse_mx6half.json").read_text()) if (DST / "r03_pmc_match_coarse_mx6half.json").exists() else None
    r = b["roofline"]
    ex = b["extra"]
    cfg = b["config"]
    nd = json.loads((DST / "r03_neardup.json").read_text())
    maps = []
    for k in nd:
        name = k.split(" | ")[0]
        if name not in maps:
            maps.append(name)
    modes = [c for c in ("int8-half", "int8", "mx6", "int8-top2", "fp16") if all(f"{m} | {c}" in nd for m in maps)]
    nd_rows = "\n".join(
        f"| {name} | {nd[name + ' | auto']['ms_per_registration']:.2f} ({nd[name + ' | auto']['pass_in_use']}"
        f"{'' if nd[name + ' | auto']['pass_in_use'] == 'fp16' else ', ' + str(nd[name + ' | auto'].get('records_in_use', '?'))}) | "
        + " | ".join(f"{nd[name + ' | ' + c]['ms_per_registration']:.2f}" for c in modes) + " | "
        f"{all(nd[name + ' | ' + c]['same_result_as_auto'] for c in modes)} | "
        f"{sum(nd[name + ' | ' + c]['fallback_queries'] for c in ['auto'] + modes)} |" for name in maps)

    def variant(k):
        v = ex.get(k)
        if not v:
            return f"`extra.{k}`: (absent)"
        rl = v.get("roofline") or {}
        which = "fp6" if rl.get("peak", 0) > 6000 else "int8"
        return (f"`extra.{k}`: **{v.get('value', float('nan')):.1f} registrations/s** ({v.get('ms_per_step', float('nan')):.3f} ms)"
                + (f", coarse kernel {rl.get('avg_launch_ms', float('nan')):.3f} ms = {rl.get('frac', float('nan')):.3f} of the {which} peak" if rl else "")
                + (f", pass in use: {v.get('coarse_pass')}" if v.get("coarse_pass") else ""))

    a6 = ex.get("A6_mutual_l2", {})
    md = f"""# Round 3 -- measurements on one MI355X (config C2: 20 000 x 200 000 x 384, 50 000 RANSAC iterations)

Produced by `bash tools/r03_final.sh` through `gpurun` (a fresh box per call; boxes of the pool differ by a few per cent),
collected by `python tools/refresh_profiles_r03.py`.  Raw files are next to this one (`r03_*`).  GPU suite on the same box:
`{text('r03_pytest_gpu.txt').splitlines()[-1] if (DST / 'r03_pytest_gpu.txt').exists() else '?'}`.

## bench.py (default: `auto` -- on D.2 descriptors the half-width pass, in fp6 since the end of round 3; operand preparation | coarse pass | two solve streams)

`python bench.py` -> `profiles/r03_bench.json`: **{b['value']:.1f} registrations/s** ({b['ms_per_step']:.3f} ms per
registration), dominant kernel `{r['kernel'].split(' (')[0]}` {r['avg_launch_ms']:.3f} ms per launch inside the timed region =
{r['achieved']:.0f} TOP/s = {r['frac']:.3f} of {r['peak'] / 1000:.1f} POP/s ({r.get('peak_note', 'dense MFMA peak')}; operations of the kernel as launched:
{r['flops_per_launch'] / 1e12:.3f} TOP -- coarse pass in use: {cfg.get('coarse_pass', '?')}; surviving chunks per query of the half-width
selection: {cfg.get('half_width_survivors_per_query')}); alone on the GPU {r['single_stream']['avg_launch_ms']:.3f} ms =
{r['single_stream']['achieved']:.0f} TOP/s = {r['single_stream']['frac']:.3f}.  `roofline.traffic` = {r.get('traffic')} bytes per launch
({r.get('traffic_source', 'PMC passes below')}).  Peak HBM allocated: {cfg.get('hbm_peak_allocated_gb', float('nan')):.1f} GB.
CPU oracle on the same box ({b['cpu_baseline']['cores']} threads): {b['cpu_baseline']['value']:.3f} registrations/s.
Pose delta vs the oracle on identical inputs (`extra.pose_delta_vs_oracle`): {ex.get('pose_delta_vs_oracle', {}).get('pose_delta_vs_oracle_frobenius')}.

The same pipeline away from the favourable case (VERDICT r2 item 2), same process, same box:

- {variant('C2_full_width')} -- `coarse="int8"` pinned: every column in the coarse pass, nothing depends on how the descriptors prune
- {variant('C2_full_width_mx6')} -- `coarse="mx6"` pinned: the same all-pairs product in microscaled fp6 on the scaled MFMA (DESIGN.md 0.8), as data independent as the line above
- {variant('C2_half_width_mx6')} -- `coarse="mx6-half"` pinned: the headline's bound in fp6 (coarse kernel 0.40 ms alone)
- {variant('C2_half_width_int8')} -- `coarse="int8-half"` pinned: the half-width pass on the int8 image (round 2's headline mode)
- {variant('C2_sustained')} -- {ex.get('C2_sustained', {}).get('steps', '?')} steps instead of 20 (the first ~15 launches after a synchronise run slower)
- {variant('C2_lifted')} -- map descriptors lifted from overlapping patch grids (near-duplicates), policy by feedback
- `extra.A6_mutual_l2`: {json.dumps(a6)[:600]}
- `extra.C3`: {ex['C3']['ms_end_to_end']:.2f} ms end to end (ViT {ex['C3']['ms_vit']:.3f}, project + lift {ex['C3']['ms_project_lift']:.3f}, registration {ex['C3']['ms_registration']:.2f}; ViT at {ex['C3']['vit_roofline']['frac']:.3f} of the fp16 MFMA peak)
- `extra.C5` (50k x 1M x 768; pass in use: {ex['C5'].get('coarse_pass', '?')}): coarse kernel {ex['C5']['ms_coarse_kernel']:.1f} ms = {ex['C5']['roofline']['frac']:.3f} of {ex['C5']['roofline']['peak'] / 1000:.0f} P(FL)OP/s, registration {ex['C5']['ms_registration']:.1f} ms; half-width pass on the int8 image, same box: {json.dumps(ex['C5'].get('int8_half_width'))}

Same box, `VFM_COARSE=int8 python bench.py` (full-width int8 pass as the whole run) -> `profiles/r03_bench_int8_full_same_box.json`:
{bi['value']:.1f} registrations/s, kernel {bi['roofline']['avg_launch_ms']:.3f} ms ({bi['roofline']['frac']:.3f} of the int8 peak).

`python bench.py --streams 1` (every kernel serialised on one stream) -> `profiles/r03_bench_streams1.json`:
{b1['value']:.1f} registrations/s, dominant kernel {b1['roofline']['avg_launch_ms']:.3f} ms.

## The timed region against what precedes it (`tools/ab_precond.sh`, `r03_warmup_ab.txt`; DESIGN.md 0.13)

```
{text('r03_warmup_ab.txt')}
```

## rocprofv3 --kernel-trace --stats of the default bench command

`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --no-cpu-baseline --no-extra`
-> `profiles/r03_bench_kernel_stats.csv` (library kernels only; 20 timed + 3 warm-up registrations + the isolated launches of
`single_stream`; the solve stages overlap the coarse pass, so a solve kernel's duration includes waiting for compute units
held by the coarse kernel):

{stats_table(DST / 'r03_bench_kernel_stats.csv', 20)}

Serial (`--streams 1`), `profiles/r03_bench_streams1_kernel_stats.csv`:

{stats_table(DST / 'r03_bench_streams1_kernel_stats.csv', 18)}

## PMC passes of the coarse kernel (`bash tools/pmc_coarse.sh`, separate --pmc passes, --kernel-trace only)

Half-width kernel (`VFM_RECORDS=3`; `profiles/r03_pmc_match_coarse_i8half.json` + `profiles/r03_pmc_half_pass*_counter_collection.csv`,
{pmh['kernel']}): {pmc_line(pmh)}.

Full-width kernel (`VFM_RECORDS=0`; `profiles/r03_pmc_match_coarse_i8.json` + `profiles/r03_pmc_pass*_counter_collection.csv`,
{pmc['kernel']}): {pmc_line(pmc)}.

fp6 kernel (`VFM_RECORDS=5`; `profiles/r03_pmc_match_coarse_mx6.json` + `profiles/r03_pmc_mx6_pass*_counter_collection.csv`,
{pm6['kernel'] if pm6 else '?'}): {pmc_line(pm6) if pm6 else '(not collected)'}.

fp6 half-width kernel (`VFM_RECORDS=7`: what the default bench runs on D.2 data since the end of round 3;
`profiles/r03_pmc_match_coarse_mx6half.json` + `profiles/r03_pmc_mx6half_pass*_counter_collection.csv`,
{pm6h['kernel'] if pm6h else '?'}): {pmc_line(pm6h) if pm6h else '(not collected)'}.

Reading: the MFMA pipe is busy ~3/4 of the cycles at a clock of ~1.75 GHz (2.4 GHz is what the 5 POP/s peak assumes): the
fraction of the peak is busy x clock / 2.4, i.e. the kernel sits against the power envelope, not against its own stalls.

## Duplicate-rich maps (`python tools/time_neardup.py`, C2 size, the bench's pipeline; `r03_neardup.json`)

ms per registration with the coarse pass chosen by the pipeline's feedback (`auto`: the pass and record kind in use after the
warm-up in brackets), and with each mode forced (int8-half = the half-width pass -- behind the device-side guard since this round,
int8 = best-score records, mx6 = the same in fp6, int8-top2 = packed top-2 records):

| map | auto | {' | '.join(modes)} | same correspondences + pose | all-pairs fallbacks |
|---|---|{'---|' * len(modes)}---|---|
{nd_rows}

## The finish stage where a query has many candidate chunks (`tools/lifted_stats.py`, `tools/prof_finish.sh`, `tools/trace_pipe.sh`)

bench.py's lifted descriptors (C2 size): rows / chunks within w of a query's best cosine (what bounds of total width w hand on),
and the library's own counts and stage times for the four full-width record kinds (`r03_lifted_stats.txt`):

```
{text('r03_lifted_stats.txt')}
```

Kernel by kernel, every kernel alone on the GPU (us; records 0 = int8 best-score, 5 = fp6 best-score; first two lines: lifted
descriptors + common component, last two: D.2; `r03_prof_finish.txt`):

```
{text('r03_prof_finish.txt')}
```

The pipeline's timeline on the same data (`auto`, overlapped), and the same kernels one after the other (`r03_lifted_cycle.txt`):

```
{text('r03_lifted_cycle.txt')}
```

## Row A6: find_correspondences' mutual L2 filter (`tools/time_pairs.py`, `tools/prof_pairs.sh`)

```
{text('r03_time_pairs.txt')}
```

Kernel sequence of one `vfm_match_mutual_pairs` call at C2 size (`r03_prof_pairs.txt`):

```
{text('r03_prof_pairs.txt')[-3000:]}
```

## F rows, C3 stages, RANSAC alone (`r03_other_rows.txt`)

```
{text('r03_other_rows.txt')}
```

C3's registration per coarse mode (`r03_time_c3_modes.txt`):

```
{text('r03_time_c3_modes.txt')}
```

ViT tile mapping A/B (`r03_ab_vit_xcd.txt`):

```
{text('r03_ab_vit_xcd.txt')[-1200:]}
```

## fp6 coarse pass (`tools/dev_mx6.py`, `tools/ab_mx6_bench.py`)

```
{text('r03_dev_mx6.txt')[-2500:]}
```

Pipelined (the bench's pipeline construction), coarse pass pinned, three kinds of data, 20 / 200 timed steps (`r03_ab_mx6_bench.txt`):

```
{text('r03_ab_mx6_bench.txt')[-4500:]}
```

## Hardware queues: the same pipeline at 1110, 1270 or 1375 registrations/s (`tools/queue_probe.py`; DESIGN.md 0.11)

Pipelines built one after the other in one process, first with private side streams (`private`), then with the side streams
shared per process (`shared`, the default since the end of round 3), int8 half-width and fp6 half-width:

```
{text('r03_queue_probe.txt')[-5000:]}
```

## What a cycle of the pipeline consists of (`tools/trace_pipe.sh`, `tools/corun_probe.py`; DESIGN.md 0.9)

```
{text('r03_pipeline_cycle.txt')[-6000:]}
```

## Soaks beyond the suite's fixed seeds

`python tools/soak_half.py 40 303`: `{text('r03_soak_half.txt').splitlines()[-1] if (DST / 'r03_soak_half.txt').exists() else '?'}`;
`python tools/soak_mx6.py 40 303`: `{text('r03_soak_mx6.txt').splitlines()[-1] if (DST / 'r03_soak_mx6.txt').exists() else '?'}`;
`python tools/soak_match.py 16 303`: `{text('r03_soak_match.txt').splitlines()[-1] if (DST / 'r03_soak_match.txt').exists() else '?'}`.

## fp6 (MX e2m3) MFMA probe (`tools/probe/mx6_probe.hip`, `r03_mx6_probe.txt`)

```
{text('r03_mx6_probe.txt')}
```
"""
    (DST / "r03_bench_summary.md").write_text(md)
    print(md[:2500])


if __name__ == "__main__":
    main()
