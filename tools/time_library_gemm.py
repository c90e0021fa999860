"""Context for the coarse pass: the vendor library GEMM (torch.mm -> hipBLASLt / rocBLAS) on the same
20 000 x 200 000 x 384 fp16 product, which has to materialise the N x M score matrix (8 GB in fp16)
and would still need a row arg-max pass over it."""
import torch
n, m, d = 20000, 200000, 384
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(n, d, device="cuda", generator=g).half()
b = torch.randn(m, d, device="cuda", generator=g).half()
out = torch.empty(n, m, device="cuda", dtype=torch.float16)
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); e.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
t = timed(lambda: torch.mm(q, b.t(), out=out))
print(f"torch.mm fp16 {n}x{d} @ {d}x{m} -> fp16 [{n}x{m}]: {t:.2f} ms = {2*n*m*d/t/1e9:.0f} TFLOP/s (writes {n*m*2/1e9:.1f} GB)")
t2 = timed(lambda: out.max(dim=1))
print(f"row max + arg-max over the materialised scores: {t2:.2f} ms")
# K-heavier shape for the library's own ceiling on this chip
a2 = torch.randn(8192, 8192, device="cuda", generator=g).half(); b2 = torch.randn(8192, 8192, device="cuda", generator=g).half()
t3 = timed(lambda: torch.mm(a2, b2))
print(f"torch.mm fp16 8192^3: {t3:.2f} ms = {2*8192**3/t3/1e9:.0f} TFLOP/s")
