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
              "prof/bench_kernel_stats.csv": "r02_bench_kernel_stats.csv",
              "prof1/bench1_kernel_stats.csv": "r02_bench_streams1_kernel_stats.csv",
              "pmc_match_coarse.json": "r02_pmc_match_coarse.json", "pytest_gpu.txt": "r02_pytest_gpu.txt",
              "neardup.json": "r02_neardup.json"}
    for i in range(1, 8):
        copies[f"pmc_pass{i}_counter_collection.csv"] = f"r02_pmc_pass{i}_counter_collection.csv"
    for a, b in copies.items():
        if (SRC / a).exists():
            shutil.copy(SRC / a, DST / b)
    b = last_json(DST / "r02_bench.json")
    b1 = last_json(DST / "r02_bench_streams1.json")
    pmc = json.loads((DST / "r02_pmc_match_coarse.json").read_text())
    r = b["roofline"]
    ex = b["extra"]
    nd = json.loads((DST / "r02_neardup.json").read_text())
    nd_rows = "\n".join(f"| {k.split(' | ')[0]} | {v['ms_per_registration']:.2f} | {v['candidate_entries_per_query']:.1f} | "
                        f"{v['coarse_records_per_query']:.1f} | {v['refined_queries']} | {v['fallback_queries']} |"
                        for k, v in nd.items() if k.endswith("pipelined"))
    md = f"""# Round 2 -- measurements on one MI355X (config C2: 20 000 x 200 000 x 384, 50 000 RANSAC iterations)

Produced by `bash tools/r02_final.sh` (GPU tests, smoke, bench plain / serial / under rocprofv3, PMC passes of the coarse
kernel) through `gpurun` (a fresh box per call; the same binary measures 352-373 registrations/s on different boxes of the
pool), collected by `python tools/refresh_profiles_r02.py`.  Raw files are next to this one (`r02_*`).

## bench.py (default: pipeline over two HIP streams, operand preparation + coarse pass | solve stage)

`python bench.py` -> `profiles/r02_bench.json`: **{b['value']:.1f} registrations/s** ({b['ms_per_step']:.3f} ms per
registration), dominant kernel {r['avg_launch_ms']:.3f} ms per launch inside the timed region =
{r['achieved']:.0f} TFLOP/s = {r['frac']:.3f} of the 2.5 PFLOP/s dense fp16 MFMA peak;
alone on the GPU {r['single_stream']['avg_launch_ms']:.3f} ms = {r['single_stream']['achieved']:.0f} TFLOP/s =
{r['single_stream']['frac']:.3f}.  CPU oracle on the same box ({b['cpu_baseline']['cores']} threads): {b['cpu_baseline']['value']:.3f} registrations/s.
Pose delta vs the oracle on identical inputs (`extra.pose_delta_vs_oracle`): {ex.get('pose_delta_vs_oracle', {}).get('pose_delta_vs_oracle_frobenius')}.
`extra.C3`: {ex['C3']['ms_end_to_end']:.2f} ms end to end (ViT {ex['C3']['ms_vit']:.2f}, project + lift {ex['C3']['ms_project_lift']:.2f}, registration {ex['C3']['ms_registration']:.2f}; ViT at {ex['C3']['vit_roofline']['frac']:.3f} of the MFMA peak).
`extra.C5`: coarse kernel {ex['C5']['ms_coarse_kernel']:.1f} ms = {ex['C5']['roofline']['frac']:.3f} of the peak, registration {ex['C5']['ms_registration']:.1f} ms.

`python bench.py --streams 1` (every kernel serialised on one stream) -> `profiles/r02_bench_streams1.json`:
{b1['value']:.1f} registrations/s, dominant kernel {b1['roofline']['avg_launch_ms']:.3f} ms.

## rocprofv3 --kernel-trace --stats of the default bench command

`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --no-cpu-baseline --no-extra`
-> `profiles/r02_bench_kernel_stats.csv` (library kernels only; 20 timed + 3 warm-up registrations + the 6 isolated
launches of `single_stream`; the solve stage overlaps the coarse pass, so a solve kernel's duration includes waiting for
compute units held by the coarse kernel):

{stats_table(DST / 'r02_bench_kernel_stats.csv')}

Serial (`--streams 1`), `profiles/r02_bench_streams1_kernel_stats.csv`:

{stats_table(DST / 'r02_bench_streams1_kernel_stats.csv', 14)}

## PMC passes of the coarse kernel (`bash tools/pmc_coarse.sh`, separate --pmc passes, --kernel-trace only)

`profiles/r02_pmc_match_coarse.json` + `profiles/r02_pmc_pass*_counter_collection.csv`: FETCH_SIZE {pmc['FETCH_SIZE_KB'] / 1024:.0f} MB (x2 per the
guide's gfx950 correction), WRITE_SIZE {pmc['WRITE_SIZE_KB'] / 1024:.0f} MB -> **{pmc['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch** (`roofline.traffic`; round 1: 1.63 GB);
L2 hit rate {pmc['TCC_hit_rate']:.3f}; clock {pmc['clock_GHz']:.2f} GHz; MFMA pipe busy {pmc['mfma_busy_fraction']:.3f} of all SIMD cycles; LDS array busy
{pmc['lds_array_busy_fraction']:.3f}; per MFMA {pmc['per_mfma']['valu_incl_mfma']:.2f} VALU (incl. the MFMA; round 1: 3.59), {pmc['per_mfma']['salu']:.2f} SALU, {pmc['per_mfma']['lds']:.2f} LDS; wave time
{pmc['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f} issuing / {pmc['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f} waiting to issue / {pmc['wave_time_shares']['SQ_WAIT_ANY']:.2f} in waitcnt + barrier.

## Duplicate-rich maps (`python tools/time_neardup.py`, C2 size, pipelined)

| map | ms / registration | candidates / query after the filter | coarse records / query | queries refined in fp32 | all-pairs fallbacks |
|---|---|---|---|---|---|
{nd_rows}

## Other evidence files

`r02_admissible_c2.json` (every C2 row is an admissible fp32 IndexFlatIP answer), `r02_c3_vit_precision.json` (fp16 ViT vs
fp32 oracle ViT: keep set, arg-max flips, pose), `r02_pmc_vit.json` + `r02_vit_kernel_stats.csv` (ViT kernels: wave lifetime
vs kernel duration), `r02_pytest_gpu.txt`.
"""
    (DST / "r02_bench_summary.md").write_text(md)
    print(md[:1500])


if __name__ == "__main__":
    main()
