"""bench.py contract on the GPU box: the plain form and the torch.distributed.run form the driver uses for
N > 1 (here with one rank: backend "nccl" = RCCL, rendezvous on 127.0.0.1), both must print ONE JSON line
with the contract's keys."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _check(out: str, steps: int):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 50 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.05 < r["frac"] < 1.0 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    assert "workload" in d["config"] and d["config"]["max_pose_err_vs_planted"] < 0.05
    return d


def test_bench_plain_small_run():
    env = dict(os.environ)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 4)
    assert d["config"]["collective"].startswith("none")
    # one pipelined warm-up step is enough: the auto policy has settled (on the half-width pass, D.2 data) before the timed region
    assert d["config"]["policy_settle_registrations"] == 6 and "half-width" in d["config"]["coarse_pass"]
    # the kernel the line reports is the kernel the C2 parity test compares with the oracle
    from tests.test_gpu_bench_config import BENCH_RECORDS_KIND
    assert d["config"]["records_kind"] == BENCH_RECORDS_KIND, d["config"]["records_kind"]


def test_bench_under_torch_distributed_run_world1():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 4)
    assert "nccl" in d["config"]["collective"]  # the RCCL process group was created and used
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    # roofline.traffic is measured in the run itself (two rocprofv3 --pmc subprocesses on the kernel the timed region ran) and is
    # what the survivor-only epilogue left of the 0.38 GB the record-writing kernel moved
    assert d["roofline"]["traffic_source"].startswith("this run"), d["roofline"]["traffic_source"]
    assert 0.05e9 < d["roofline"]["traffic"] < 0.25e9, d["roofline"]["traffic"]
    # the metric's "pose delta vs ref" and the other configs ride on the same line, outside the timed region
    ex = d["extra"]
    assert "error" not in ex, ex
    if "pose_delta_vs_oracle" in ex:   # present whenever the host finishes the whole oracle registration in the time budget
        assert ex["pose_delta_vs_oracle"]["pose_delta_vs_oracle_frobenius"] <= 1e-5
    # figures that do not rest on D.2's prunable noise: full-width pass, 200-step run, lifted-like descriptors
    assert "error_c2_variants" not in ex, ex
    fw = ex["C2_full_width"]
    assert fw["coarse_pass"] == "int8, best-score records" and fw["roofline"]["columns_multiplied"] == 384
    assert fw["value"] > 100 and 0.05 < fw["roofline"]["frac"] < 1.0
    assert abs(fw["roofline"]["flops_per_launch"] / (fw["roofline"]["avg_launch_ms"] * 1e-3) / 1e12 / fw["roofline"]["peak"] - fw["roofline"]["frac"]) < 1e-9
    for k in ("C2_full_width_mx6", "C2_full_width_mx6_fused", "C2_half_width_mx6"):   # the fp6 forms of the two lines above: same construction, fp6 roofline
        assert ex[k]["value"] > 100 and ex[k]["roofline"]["peak"] == 10000.0 and 0.05 < ex[k]["roofline"]["frac"] < 1.0, k
        assert "fp6" in ex[k]["coarse_pass"], k
    assert ex["C2_sustained"]["steps"] == 200 and ex["C2_sustained"]["value"] > 100
    lf = ex["C2_lifted"]
    assert lf["value"] > 100 and lf["pose_err_vs_planted"] < 0.05 and lf["correspondences"] > 5000
    assert 0 < ex["A6_mutual_l2"]["ms_mutual_pairs"] < 6.0 and ex["A6_mutual_l2"]["mutual_pairs"] > 9000
    assert ex["C3"]["ms_end_to_end"] > ex["C3"]["ms_vit"] > 0 and 0 < ex["C3"]["vit_roofline"]["frac"] < 1
    assert ex["C3_pipelined"]["value"] > 100 and ex["C3_pipelined"]["grouped"]["value"] > 100 and ex["C3_pipelined"]["grouped"]["pairs_per_vit_call"] == 7
    assert ex["C5"]["pose_err_vs_planted"] < 0.05 and 0.05 < ex["C5"]["roofline"]["frac"] < 1
    f16 = ex["C5"]["fp16_descriptor_storage"]   # configs[4]: the map stored in fp16 -- half the bytes, the registration of the widened rows
    assert "error" not in f16 and f16["pose_equals_the_widened_rows_registration"] and f16["map_bytes"] * 2 == f16["map_bytes_fp32"]
    assert f16["pose_err_vs_planted"] < 0.05 and 0 < f16["ms_registration"] < 40
    # every stage alone on the GPU with its algorithmic work and the fraction of the peak that bounds it
    stg = ex["stages"]
    assert "error" not in stg and "error_c3_rows" not in stg, stg
    assert 0.05 < stg["coarse pass"]["frac"] < 1.0 and stg["coarse pass"]["bound"] == "mfma"
    prep = stg["prepare (normalise rows, int8 + fp6 images)"]
    assert prep["bound"] == "hbm" and 0.1 < prep["frac"] < 1.0 and 0.05 < prep["ms"] < 0.5
    assert stg["RANSAC + Kabsch (50 000 hypotheses, fp64)"]["correspondences"] > 5000
    assert 0 < stg["ViT-S/14 on 6 x 1200x1600 (C3)"]["frac"] < 1 and 0 < stg["projection + lifting, 6 cameras (C3)"]["frac"] < 1
    assert abs(stg["sum_of_stages_ms"] - sum(v["ms"] for k, v in stg.items() if isinstance(v, dict) and "(C3)" not in k)) < 1e-6


def test_bench_pairs_form_for_config_c4():
    """`--pairs P`: pair p is generated from seed 42 + p on rank p mod N and registered there; per-rank rates are reported."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", str(ROOT / "bench.py"), "--gpus", "1", "--pairs", "32", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 32)
    assert d["config"]["scene_pairs_total"] == 32 and d["config"]["resident_scene_pairs_per_gpu"] == 32
    assert len(d["per_rank_registrations_per_s"]) == 1 and d["per_rank_registrations_per_s"][0] >= d["value"] * 0.99
    assert "[rank 0] 32 registrations" in r.stderr
    # config C4 puts 32 resident pairs + the pipeline's three buffer sets on every GPU: far below the 288 GB of one MI355X,
    # so an 8-GPU run cannot run out of memory on first contact
    assert 5.0 < d["config"]["hbm_peak_allocated_gb"] < 40.0


@pytest.mark.parametrize("group", [1, 4])
def test_bench_c3_form_shards_end_to_end_pairs(group):
    """`--form c3 --pairs P`: the job's pairs end to end from uint8 images, sharded like the c2 form; the planted
    transform (the identity) is recovered for every pair -- one ViT call per pair, and the cameras of four pairs per call
    (`--feature-group 4`: 10 pairs = groups of 4 + 4 + 2)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29523 + group), str(ROOT / "bench.py"), "--gpus", "1", "--form", "c3", "--pairs", "10", "--warmup", "1",
           "--feature-group", str(group)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["config"]["scene_pairs_total"] == 10 and d["unit"] == "registrations/s"
    assert d["value"] > 50 and d["config"]["max_pose_err_vs_planted"] < 0.05 and d["config"]["correspondences_last_step"] > 15000
    assert "C3" in d["metric"] and d["roofline"] is None and d["config"]["pairs_per_vit_call"] == group
