"""Exercise the RCCL code path of vfmreg.dist on ONE GPU (world_size 1 process group)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
import torch.distributed as dist

from vfmreg import dist as vd

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
poses = torch.randn(5, 4, 4, dtype=torch.float64, device=dev)
aux = torch.arange(5, device=dev)
T, a = vd.gather_poses(poses, aux, 5, 0, 1)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(T, poses) and torch.equal(a, aux) and t.item() == 1.5
print("rccl world-1 path ok:", dist.get_backend())
dist.destroy_process_group()
