"""Multi-GPU execution of the path.

The reference has no distributed code.  A registration reads only its own scan/map pair
(registration_node.py:587-589 iterates scans independently), so the path shards by INDEPENDENT SCENE
PAIRS: one process per GPU (torchrun), pair p runs on rank p mod G, no data-path collective.  The
only exchange is the final gather of the 4x4 poses (+ inlier counts): one
``all_gather_into_tensor`` of fp64[pairs_per_rank, 4, 4] -- backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests.  256 pairs x 128 B = 32 KB: latency-bound, topology irrelevant.

Second mode (optional: one pair that must beat one GPU, e.g. config C5): the MAP's rows are split over the ranks,
every rank searches its shard for all queries, and one ``all_reduce(MAX)`` over packed (similarity, row) keys -- N x 8 bytes --
leaves the global top-1 of every query on every rank (``shard_map_rows`` / ``reduce_top1``).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world).
    Under a launcher (RANK and WORLD_SIZE set) the group is created even for one rank, so that a
    single-GPU run exercises the same RCCL calls as an 8-GPU run."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def _host_staged(t: torch.Tensor) -> bool:
    """True where the collective has to go through host memory: the "gloo" backend with device tensors (the form the single-GPU
    two-rank test of the real path uses -- RCCL refuses two ranks on one device, gloo does not care where the ranks compute)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_gather_rows(out: torch.Tensor, buf: torch.Tensor) -> None:
    """``all_gather_into_tensor(out, buf)`` on whatever backend the group has (RCCL over xGMI: directly on the device tensors)."""
    if _host_staged(buf):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, buf.cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, buf)


def all_reduce_max(t: torch.Tensor) -> None:
    """In-place ``all_reduce(MAX)``, host-staged under gloo like ``all_gather_rows``."""
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def shard_pairs(num_pairs: int, rank: int, world: int) -> List[int]:
    """Pair p -> rank p mod world (round robin)."""
    return list(range(rank, num_pairs, world))


def pairs_per_rank(num_pairs: int, world: int) -> int:
    return (num_pairs + world - 1) // world


def gather_poses(local_poses: torch.Tensor, local_aux: torch.Tensor, num_pairs: int, rank: int, world: int
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the per-rank results and restore global pair order.

    local_poses: [n_local, 4, 4] fp64, local_aux: [n_local] int64 (e.g. correspondence counts), for
    the pairs of ``shard_pairs`` in order.  Returns ([num_pairs, 4, 4], [num_pairs]) on every rank.
    """
    if world == 1 and local_poses.shape[0] == num_pairs:
        # one rank owns every pair, in order: nothing to gather or to reorder
        return local_poses.clone(), local_aux.to(torch.int64).clone()
    cap = pairs_per_rank(num_pairs, world)
    dev = local_poses.device
    buf = torch.zeros((cap, 17), dtype=torch.float64, device=dev)
    n_local = local_poses.shape[0]
    if n_local:
        buf[:n_local, :16] = local_poses.reshape(n_local, 16)
        buf[:n_local, 16] = local_aux.to(torch.float64)
    if dist.is_available() and dist.is_initialized():
        out = torch.empty((world, cap, 17), dtype=torch.float64, device=dev)
        all_gather_rows(out.view(world * cap, 17), buf)  # RCCL over xGMI on GPUs, gloo on CPU
    else:
        assert world == 1, "world > 1 needs an initialised process group"
        out = buf.unsqueeze(0)
    poses = torch.empty((num_pairs, 4, 4), dtype=torch.float64, device=dev)
    aux = torch.empty(num_pairs, dtype=torch.int64, device=dev)
    for r in range(world):
        ids = shard_pairs(num_pairs, r, world)
        if ids:
            idx = torch.tensor(ids, device=dev)
            poses[idx] = out[r, :len(ids), :16].reshape(len(ids), 4, 4)
            aux[idx] = out[r, :len(ids), 16].round().to(torch.int64)
    return poses, aux


def register_sharded(num_pairs: int, register_pair: Callable[[int], Tuple[torch.Tensor, torch.Tensor]],
                     rank: int, world: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run ``register_pair(p) -> (T[4,4] fp64, count int64[1][, ready])`` for this rank's pairs (enqueue
    only, no host sync), then gather.  ``ready`` (optional) is the ``torch.cuda.Event`` or ``torch.cuda.Stream`` after
    which T and count are complete -- e.g. ``out["done"]`` / ``out["result_stream"]`` of a
    ``RegistrationPipeline(overlap_ransac=True)``, whose results are produced on a side stream; without it the
    results must be ordered on the caller's current stream.  Config C4: 256 pairs over 8 GPUs."""
    ids = shard_pairs(num_pairs, rank, world)
    poses = torch.zeros((len(ids), 4, 4), dtype=torch.float64, device=device)
    aux = torch.zeros(len(ids), dtype=torch.int64, device=device)
    for j, p in enumerate(ids):
        res = register_pair(p)
        T, cnt = res[0], res[1]
        ready = res[2] if len(res) > 2 else None
        if ready is not None and T.is_cuda:
            cur = torch.cuda.current_stream(T.device)
            if isinstance(ready, torch.cuda.Event):
                cur.wait_event(ready)
            else:
                cur.wait_stream(ready)
        poses[j].copy_(T)
        aux[j:j + 1].copy_(cnt.reshape(1))
    return gather_poses(poses, aux, num_pairs, rank, world)


def shard_map_rows(m: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of a map of m rows for this rank (whole 128-row chunks of the matcher except at the end)."""
    per = ((m + world - 1) // world + 127) // 128 * 128
    lo = min(m, rank * per)
    return lo, min(m, lo + per)


def pack_top1(idx_local: torch.Tensor, sim: torch.Tensor, row_offset: int) -> torch.Tensor:
    """(index into this rank's shard, similarity) of every query -> one non-negative int64 key per query whose order is
    "higher similarity first, then LOWER global row" (the oracle's tie rule): 32 bits of the similarity's order-preserving
    integer image above 31 bits of (2^31 - 1 - global row).  Unresolved queries (index < 0: the gated search proved them below
    the caller's threshold) pack as 0, below every resolved one."""
    bits = sim.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    neg = bits >= 0x80000000
    key32 = torch.where(neg, 0xFFFFFFFF - bits, bits + 0x80000000)          # float order -> unsigned order
    row = idx_local.to(torch.int64) + int(row_offset)
    packed = (key32 << 31) | (0x7FFFFFFF - row)
    return torch.where(idx_local >= 0, packed, torch.zeros_like(packed))


def unpack_top1(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of ``pack_top1``: (global row int64, similarity fp32); key 0 -> (-1, -2.0), the gated search's "no match"."""
    key32 = packed >> 31
    bits = torch.where(key32 >= 0x80000000, key32 - 0x80000000, 0xFFFFFFFF - key32)
    sim = (bits & 0xFFFFFFFF).to(torch.int64)
    sim = torch.where(sim >= 0x80000000, sim - (1 << 32), sim).to(torch.int32).view(torch.float32)
    row = 0x7FFFFFFF - (packed & 0x7FFFFFFF)
    none = packed == 0
    return torch.where(none, torch.full_like(row, -1), row), torch.where(none, torch.full_like(sim, -2.0), sim)


def reduce_top1(idx_local: torch.Tensor, sim: torch.Tensor, row_offset: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Global top-1 per query from every rank's top-1 over its own map rows: one ``all_reduce(MAX)`` of N int64 keys (RCCL over
    xGMI on GPUs: 160 KB at N = 20 000, 400 KB at C5's 50 000 -- microseconds).  Every rank returns the same (row, similarity);
    identical to the unsharded search wherever that resolves the query (the similarity is the fp32 image of the exact fp64
    score of the winning row on whichever rank holds it; ties go to the lower global row)."""
    packed = pack_top1(idx_local, sim, row_offset)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        all_reduce_max(packed)
    return unpack_top1(packed)
