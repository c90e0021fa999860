import os, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0)) if False else dist.init_process_group("nccl")
x = torch.zeros(1, device="cuda")
for name, fn in (("dist.barrier()", lambda: dist.barrier()), ("all_reduce(1) + synchronize", lambda: (dist.all_reduce(x), torch.cuda.synchronize())),
                 ("dist.barrier(device_ids=[0])", lambda: dist.barrier(device_ids=[0]))):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); print(f"{name}: median {ts[10]:.3f} ms, min {ts[0]:.3f}, max {ts[-1]:.3f}")
dist.destroy_process_group()
