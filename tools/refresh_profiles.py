"""Copy the artefacts of `bash tools/full_run.sh` (gpurun_out/r01final) and `bash tools/pmc_coarse.sh`
(gpurun_out/pmc_coarse) into profiles/ and regenerate profiles/r01_bench_summary.md."""
import csv
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC, PMC, DST = ROOT / "gpurun_out" / "r01final", ROOT / "gpurun_out" / "pmc_coarse", ROOT / "profiles"
shutil.copy(SRC / "prof" / "bench_kernel_stats.csv", DST / "r01_bench_kernel_stats.csv")
shutil.copy(SRC / "prof1" / "bench1_kernel_stats.csv", DST / "r01_bench_streams1_kernel_stats.csv")
(DST / "r01_bench.json").write_text((SRC / "bench.json").read_text().strip().splitlines()[-1] + "\n")
(DST / "r01_bench_streams1.json").write_text((SRC / "bench_prof1.json").read_text().strip().splitlines()[-1] + "\n")
shutil.copy(SRC / "pytest_gpu.txt", DST / "r01_pytest_gpu.txt")
if (ROOT / "gpurun_out" / "other_rows.txt").exists():
    shutil.copy(ROOT / "gpurun_out" / "other_rows.txt", DST / "r01_other_rows.txt")
if (PMC / "pmc_match_coarse.json").exists():
    shutil.copy(PMC / "pmc_match_coarse.json", DST / "r01_pmc_match_coarse.json")
    for i in range(1, 8):
        shutil.copy(PMC / f"p{i}_counter_collection.csv", DST / f"r01_pmc_pass{i}_counter_collection.csv")


def table(fn, n=18):
    out = ["| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in list(csv.DictReader(open(fn)))[:n]:
        if "(anonymous namespace)" not in r["Name"] and "_GLOBAL__N_" not in r["Name"]:
            continue  # torch / runtime kernels of the input generation
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out)


b = json.loads((DST / "r01_bench.json").read_text())
b1 = json.loads((DST / "r01_bench_streams1.json").read_text())
pmc = json.loads((DST / "r01_pmc_match_coarse.json").read_text())
ntests = (DST / "r01_pytest_gpu.txt").read_text().strip().splitlines()[-1]
md = f"""# Round 1 -- measurements on one MI355X (config C2: 20 000 x 200 000 x 384, 50 000 RANSAC iterations)

Produced by `bash tools/full_run.sh` and `bash tools/pmc_coarse.sh` through `gpurun` (a fresh box per call; the
same binary measures 335-361 registrations/s and 2.51-2.75 ms for the dominant kernel on different boxes of the
pool), collected by `python tools/refresh_profiles.py`.  Raw files are next to this one.

## bench.py (default: pipeline over three HIP streams)

`python bench.py` -> `profiles/r01_bench.json`: **{b['value']:.1f} registrations/s** ({b['ms_per_step']:.3f} ms per
registration), dominant kernel {b['roofline']['avg_launch_ms']:.3f} ms per launch under overlap =
{b['roofline']['achieved']:.0f} TFLOP/s = {b['roofline']['frac']:.3f} of the 2.5 PFLOP/s dense fp16 MFMA peak;
alone on the GPU {b['roofline']['single_stream']['avg_launch_ms']:.3f} ms = {b['roofline']['single_stream']['achieved']:.0f} TFLOP/s =
{b['roofline']['single_stream']['frac']:.3f}.  CPU oracle on the same box ({b['cpu_baseline']['cores']} threads): {b['cpu_baseline']['value']:.3f} registrations/s.

`python bench.py --streams 1` (every kernel serialised on one stream) -> `profiles/r01_bench_streams1.json`:
{b1['value']:.1f} registrations/s, dominant kernel {b1['roofline']['avg_launch_ms']:.3f} ms.

## rocprofv3 --kernel-trace --stats of the default bench command

`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --no-cpu-baseline`
-> `profiles/r01_bench_kernel_stats.csv` (library kernels only; 20 timed + 3 warm-up registrations + the 6 isolated
launches of `single_stream`; the stages overlap, so a side-stage kernel's duration includes waiting for compute units
held by the coarse kernel):

{table(DST / 'r01_bench_kernel_stats.csv', 24)}

Same with `--streams 1` (no overlap: per-kernel times are the undisturbed ones) -> `profiles/r01_bench_streams1_kernel_stats.csv`:

{table(DST / 'r01_bench_streams1_kernel_stats.csv', 24)}

The HIP-event average `bench.py` reports for the dominant kernel (`roofline.avg_launch_ms`) and the rocprofv3
average of the same command (first table row; it also contains the 6 undisturbed `single_stream` launches) agree.

## PMC passes for the dominant kernel (`tools/pmc_coarse.sh`: 7 separate --pmc passes, --kernel-trace only)

`profiles/r01_pmc_match_coarse.json`, raw `profiles/r01_pmc_pass*_counter_collection.csv`:

* HBM traffic per launch: 2 x FETCH_SIZE + WRITE_SIZE = {pmc['hbm_bytes_per_launch'] / 1e9:.2f} GB (algorithmic: 0.17 GB fp16 operands +
  0.25 GB partial records); L2 hit rate {pmc['TCC_hit_rate']:.3f}.  {pmc['hbm_bytes_per_launch'] / 1e9 / (pmc['median_duration_us_under_pmc'] * 1e-6) / 1e3:.2f} TB/s: far from the HBM roof.
* clock under the kernel {pmc['clock_GHz']:.2f} GHz (power-limited; 2.4 GHz nominal), MFMA pipe busy {pmc['mfma_busy_fraction']:.3f},
  LDS array busy {pmc['lds_array_busy_fraction']:.3f}, LDS bank conflicts {pmc['counters_avg_per_launch']['SQ_LDS_BANK_CONFLICT']:.0f}.
* wave time: issuing {pmc['wave_time_shares']['SQ_ACTIVE_INST_ANY']:.2f}, waiting to issue {pmc['wave_time_shares']['SQ_WAIT_INST_ANY']:.2f},
  s_waitcnt / barrier {pmc['wave_time_shares']['SQ_WAIT_ANY']:.2f} (of which LDS {pmc['wave_time_shares']['SQ_WAIT_INST_LDS']:.3f}).
* per MFMA: {pmc['per_mfma']['lds']:.2f} LDS instructions, {pmc['per_mfma']['valu_incl_mfma'] - 1:.2f} other VALU, {pmc['per_mfma']['salu']:.2f} SALU.

Ablations of the same kernel (timing only; `tools/build_ablate.sh` + `tools/ablate.py`, previous kernel generation, 2.74 ms
baseline): no fold 2.69 ms, no LDS-DMA 2.42 ms, no LDS fragment reads 2.47 ms, no fold + no DMA 2.31 ms, MFMA + barrier
skeleton only 1.98 ms (1.55 PFLOP/s at 1.66 GHz, MFMA busy 0.88) -- the power-limited ceiling of this tile structure.

## Other rows (tools/time_*.py via tools/other_rows.sh; raw output: profiles/r01_other_rows.txt)

C5 coarse pass 50 000 x 1 000 000 x 768: 72.4 ms (1.06 PFLOP/s).  Mutual Euclidean NN (A6) 20k x 200k x 384: 17.4 ms.
ViT-S/14 on 6 x 1200x1600: 0.83 ms; 6-camera lift of 20 000 points: 0.07 ms; C3 one pair end to end: 4.21 ms
(`profiles/r01_c3_features_kernel_stats.csv`).  Real-data regime (1500 x 100k): 0.33 ms per registration.

## GPU test-suite

`python -m pytest tests -m gpu -q` -> `profiles/r01_pytest_gpu.txt`: {ntests}
"""
(DST / "r01_bench_summary.md").write_text(md)
print(md[:1500])
