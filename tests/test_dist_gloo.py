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


def _shard_worker(rank, world, port, out_dir):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "vfm-registration_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vfmreg import dist as vd
    vd.init_from_env(backend="gloo")
    q, b = _shard_case()
    lo, hi = vd.shard_map_rows(b.shape[0], rank, world)
    s = q @ b[lo:hi].T                      # stand-in for the search of this rank's rows (the HIP path needs a GPU)
    sim, idx = s.max(dim=1)
    first = (s == sim[:, None]).to(torch.int64).argmax(dim=1)   # ties -> lowest row, as the matcher decides them
    gate = 0.95
    idx_l = torch.where(sim >= gate, first, torch.full_like(first, -1))   # the gated search leaves the rest unresolved
    sim_l = torch.where(sim >= gate, sim, torch.full_like(sim, -2.0))
    gi, gs = vd.reduce_top1(idx_l, sim_l, lo)
    torch.save((gi, gs), os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _shard_case():
    g = torch.Generator().manual_seed(5)
    b = torch.nn.functional.normalize(torch.randn(1000, 16, generator=g), dim=1)
    b[700] = b[3]                # an exact duplicate across the shard boundary: the lower row must win
    q = torch.cat([b[[3, 650, 999]], torch.nn.functional.normalize(torch.randn(61, 16, generator=g), dim=1)])
    q[10] = -b[5]                # a query far from its best row
    return q.float(), b.float()


def test_map_row_sharding_world2(tmp_path):
    """SURVEY.md 8 E, second mode: the map's rows over two ranks, one all_reduce(MAX) of packed (similarity, row) keys."""
    port = _free_port()
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    q, b = _shard_case()
    s = q @ b.T
    sim = s.max(dim=1).values
    first = (s == sim[:, None]).to(torch.int64).argmax(dim=1)
    want_i = torch.where(sim >= 0.95, first, torch.full_like(first, -1))
    want_s = torch.where(sim >= 0.95, sim, torch.full_like(sim, -2.0))
    assert want_i[:3].tolist() == [3, 650, 999] and int((want_i < 0).sum()) > 0
    for r in range(2):
        gi, gs = torch.load(os.path.join(tmp_path, f"s{r}.pt"))
        assert torch.equal(gi, want_i) and torch.equal(gs, want_s)


def test_pack_top1_order_and_round_trip():
    from vfmreg import dist as vd
    sims = torch.tensor([-2.0, -1.0, -1e-30, -0.0, 0.0, 1e-30, 0.5, 0.7999999, 0.8, 1.0], dtype=torch.float32)
    idx = torch.arange(10)
    k = vd.pack_top1(idx, sims, 100)
    assert bool((k > 0).all()) and bool((k[1:] >= k[:-1]).all())          # order of the similarities (-0.0 / 0.0 next to each other)
    gi, gs = vd.unpack_top1(k)
    assert torch.equal(gi, idx + 100) and torch.equal(gs.view(torch.int32), sims.view(torch.int32))
    same = vd.pack_top1(torch.tensor([7, 2]), torch.tensor([0.9, 0.9]), 0)
    assert int(same[1]) > int(same[0])                                    # equal similarity: the lower row wins the MAX
    gi, gs = vd.unpack_top1(vd.pack_top1(torch.tensor([-1, 4]), torch.tensor([-2.0, 0.3]), 50))
    assert gi.tolist() == [-1, 54] and gs.tolist()[0] == -2.0
