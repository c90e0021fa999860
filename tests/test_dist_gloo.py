"""Multi-process tests of the scene-pair sharding + pose gather on CPU:
world_size 2, gloo backend, rendezvous on 127.0.0.1.  The per-pair work is a deterministic stand-in
(the HIP path needs a GPU); what is tested is the partitioning, ordering and the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_pose(p):
    rng = np.random.default_rng(1000 + p)
    T = np.eye(4)
    T[:3, :] = rng.standard_normal((3, 4))
    return torch.from_numpy(T), torch.tensor([p * 7 + 3], dtype=torch.int64)


def _worker(rank, world, port, num_pairs, out_dir):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "vfm-registration_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vfmreg import dist as vd
    r, w = vd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    poses, aux = vd.register_sharded(num_pairs, _fake_pose, rank, world, torch.device("cpu"))
    torch.save((poses, aux), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [8, 5, 1])
def test_pair_sharding_and_gather_world2(tmp_path, num_pairs):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_pairs, str(tmp_path)), nprocs=2, join=True)
    want_T = torch.stack([_fake_pose(p)[0] for p in range(num_pairs)])
    want_a = torch.tensor([p * 7 + 3 for p in range(num_pairs)])
    for r in range(2):
        poses, aux = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(poses, want_T) and torch.equal(aux, want_a)  # every rank holds all poses, in pair order


def test_shard_pairs_partition():
    from vfmreg import dist as vd
    for world in (1, 2, 4, 8):
        seen = sorted(p for r in range(world) for p in vd.shard_pairs(256, r, world))
        assert seen == list(range(256))
        assert all(len(vd.shard_pairs(256, r, world)) == 256 // world for r in range(world))
    assert vd.shard_pairs(5, 1, 2) == [1, 3] and vd.pairs_per_rank(5, 2) == 3
    T, a = vd.gather_poses(torch.eye(4, dtype=torch.float64)[None], torch.tensor([9]), 1, 0, 1)
    assert torch.equal(T[0], torch.eye(4, dtype=torch.float64)) and a.tolist() == [9]
