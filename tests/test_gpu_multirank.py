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
