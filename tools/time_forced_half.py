"""A pipeline pinned to the half-width pass on data it does not prune (the device-side guard's cost): descriptors that are all alike,
lifted descriptors with and without a common component; int8 and fp6 forms; ms per C2-size registration, serial."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import time
import torch
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
n, m, d = 20000, 200000, 384
data = {"lifted": synth.make_lifted_pair_device(n, m, d, seed=42, view_noise=0.1),
        "lifted + common": synth.make_lifted_pair_device(n, m, d, seed=42, view_noise=0.1, common=1.0)}
for name, p in data.items():
    ref = None
    for mode in ("int8-top2", "int8-half", "mx6-half"):
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, coarse=mode)
        for _ in range(2):
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        k = int(out["count"].item())
        sig = (k, out["corres"][:k].clone(), out["T"].clone())
        same = ref is None or (sig[0] == ref[0] and torch.equal(sig[1], ref[1]) and torch.equal(sig[2], ref[2]))
        ref = ref or sig
        print(f"{name:16s} {mode:10s}: {ms:7.2f} ms per registration, {k} correspondences, same result as int8-top2: {same}", flush=True)
        del pipe
