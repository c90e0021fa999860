"""Row E on ONE GPU (VERDICT r4 "next" item 6): the real `bench.py` path -- HIP kernels, resident pairs, `shard_pairs`,
`gather_poses` with ragged padding -- driven by `torch.distributed.run` with TWO ranks that share device 0.  RCCL refuses two
ranks on one device, so the process group is gloo (`--backend gloo`: the collectives are staged through host memory, everything
else is the N-GPU code path); the same job is then run by one process, and every gathered pose / correspondence count of the
two-rank job must be bit-equal to the single-process pose of the same global pair (pair p is generated from seed 42 + p on
whichever rank owns it).  Both forms of the job: descriptors resident (c2) and end to end from images (c3)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(tmp_path, tag, world, extra, port):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dump = tmp_path / f"poses_{tag}.npz"
    tail = [str(ROOT / "bench.py"), "--gpus", str(world), "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--backend", "gloo",
            "--device-index", "0", "--dump-poses", str(dump)] + extra
    if world == 1:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]       # rank 0 prints the one line of the job
    z = np.load(dump)
    return json.loads(lines[0]), z["poses"], z["counts"], r.stderr


@pytest.mark.parametrize("form", ["c2", "c3"])
def test_two_ranks_on_one_device_equal_the_single_process_job(tmp_path, form):
    pairs = 7                                       # ragged: rank 0 owns pairs 0 2 4 6, rank 1 owns 1 3 5 (padded to 4 rows in the gather)
    extra = ["--pairs", str(pairs)] + (["--form", "c3", "--scan-rows", "6000", "--map-rows", "40000", "--iters", "5000"] if form == "c3" else [])
    d2, poses2, counts2, err2 = _run(tmp_path, form + "_w2", 2, extra, 29541 if form == "c2" else 29543)
    d1, poses1, counts1, _ = _run(tmp_path, form + "_w1", 1, extra, 0)
    assert d2["n_gpus"] == 2 and d2["steps"] == 4 and d2["config"]["scene_pairs_total"] == pairs and d1["steps"] == pairs
    assert "gloo" in d2["config"]["collective"]
    assert len(d2["per_rank_registrations_per_s"]) == 2 and all(x > 10 for x in d2["per_rank_registrations_per_s"])
    if form == "c2":
        assert "[rank 0] 4 registrations" in err2 and "[rank 1] 3 registrations" in err2
    assert poses2.shape == (pairs, 4, 4) and counts2.shape == (pairs,)
    assert (counts2 > 1000).all()
    np.testing.assert_array_equal(counts2, counts1)
    np.testing.assert_array_equal(poses2, poses1)   # bit-equal, pair by pair, whichever rank registered it
    assert d2["config"]["max_pose_err_vs_planted"] < 0.05


def test_c4_workload_256_pairs_over_two_ranks(tmp_path):
    """BASELINE.json configs[3] -- 256 independent scene pairs at C2's own size (20 000 x 200 000 x 384, 50 000 RANSAC iterations),
    sharded pair p -> rank p mod N, one gather of the poses -- over two ranks that share the device (the 8-GPU /
    RCCL execution is the driver's).  Every rank keeps `resident_scene_pairs_per_gpu` = 32 of its 128 pairs in HBM and cycles through
    them: its pair j is registered on the data of its pair j mod 32.  All 256 gathered poses and correspondence counts, in global pair
    order, against a single-process job that holds 64 DISTINCT pairs (no wrap-around there): pose[p] must be bit-equal to the
    single-process pose of the pair whose data the owning rank actually registered, (p mod 2) + 2 ((p div 2) mod 32)."""
    pairs, world, resident = 256, 2, 32
    d2, poses2, counts2, err2 = _run(tmp_path, "c4_w2", world, ["--pairs", str(pairs), "--resident", str(resident)], 29547)
    d1, poses1, counts1, _ = _run(tmp_path, "c4_w1", 1, ["--pairs", str(world * resident), "--resident", str(world * resident)], 0)
    assert d2["n_gpus"] == 2 and d2["steps"] == pairs // world and d2["config"]["scene_pairs_total"] == pairs
    assert d2["config"]["resident_scene_pairs_per_gpu"] == resident and d1["config"]["resident_scene_pairs_per_gpu"] == world * resident
    assert "[rank 0] 128 registrations" in err2 and "[rank 1] 128 registrations" in err2
    assert poses2.shape == (pairs, 4, 4) and counts2.shape == (pairs,) and poses1.shape == (world * resident, 4, 4)
    src = np.array([(p % world) + world * ((p // world) % resident) for p in range(pairs)])
    assert len(set(src.tolist())) == world * resident          # every resident pair of both ranks is registered (four times)
    np.testing.assert_array_equal(counts2, counts1[src])
    np.testing.assert_array_equal(poses2, poses1[src])          # bit-equal, all 256, in global pair order
    assert (counts2 > 1000).all()
    # the 64 distinct pairs have 64 distinct poses: an ordering mistake in the gather could not hide behind equal rows
    assert len({poses1[k].tobytes() for k in range(world * resident)}) == world * resident
    assert d2["config"]["max_pose_err_vs_planted"] < 0.05
