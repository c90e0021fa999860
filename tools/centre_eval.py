#!/usr/bin/env python
"""VERDICT r5 item 1, evaluated before anything is built: does CENTRING the operands in front of the fp6 quantisation shrink what the
full-width fp6 pass hands to the finish stage on descriptors that are alike?

    <q, b> = <q - nu, b - mu> + <nu, b - mu> + <q, mu>          (any fixed nu, mu: the arg-max over b and the exact decision are untouched)

For each data set (bench.py's C2_lifted with and without the common component, C3's map = 90 % Gaussian + 10 % lifted rows of the
ViT's own features) and each variant -- plain, map centred (nu = 0: no per-row bias in the coarse kernel), both centred -- this
script emulates the MX fp6 (e2m3, one power-of-two scale per 32 columns, round to nearest even) image in torch, MEASURES the residuals
E = |v - dequantised v| as the preparation does, and counts the candidate chunks per query of match_select_best_kernel's rule:
a chunk is a candidate iff  best + bound >= max(lower bound of the query's best chunk, gate_q),  bound = E_q rho_c + (|q~| + E_q) E_c.
Also: the half-width survivors per query (x_half + bound_half + r_q R_c >= gate_q) for the same variants.
Runs on the GPU through torch only (no library kernels): python tools/centre_eval.py [n_queries]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import synth  # noqa: E402

dev = "cuda"
NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 4000


def fp6_quant(v: torch.Tensor) -> torch.Tensor:
    """MX e2m3 image of rows v [r, d] (d % 32 == 0), dequantised: scale 2^e per 32-column block with max / 2^e <= 7.5"""
    r, d = v.shape
    x = v.view(r, d // 32, 32)
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 7.5))
    s = torch.exp2(e)
    y = (x / s).clamp(-7.5, 7.5)
    a = y.abs()
    # e2m3: steps 0.125 below 2, 0.25 in [2, 4), 0.5 in [4, 7.5]
    step = torch.where(a < 2.0, 0.125, torch.where(a < 4.0, 0.25, 0.5)).to(v.dtype)
    qv = torch.round(a / step) * step     # (torch.round: half to even)
    return (torch.sign(y) * qv * s).view(r, d)


def evaluate(name, q, b, gate=0.8):
    n, d = q.shape
    m = b.shape[0]
    qn = torch.nn.functional.normalize(q.double(), dim=1).float()
    bn = torch.nn.functional.normalize(b.double(), dim=1).float()
    mp = (m + 127) // 128 * 128
    mu_map = bn.mean(0)
    mu_scan = qn.mean(0)
    print(f"== {name}: n {n}, m {m}; |mean of map rows| {mu_map.norm():.3f}, |mean of scan rows| {mu_scan.norm():.3f}")
    variants = [("plain", None, None), ("map centred (mu = map mean)", None, mu_map), ("both centred (mu = nu = map mean)", mu_map, mu_map),
                ("both centred (mu = nu = scan mean)", mu_scan, mu_scan), ("both (mu = map mean, nu = scan mean)", mu_scan, mu_map)]
    for vname, nu, mu in variants:
        qt = qn - nu if nu is not None else qn
        bt = bn - mu if mu is not None else bn
        qh, bh = fp6_quant(qt), fp6_quant(bt)
        Eq = (qt - qh).norm(dim=1)
        Eb = (bt - bh).norm(dim=1)
        nq, nb = qt.norm(dim=1), bt.norm(dim=1)
        alpha = (qn @ mu) if mu is not None else torch.zeros(n, device=dev)          # <q, mu>
        beta = (bt @ nu) if nu is not None else torch.zeros(m, device=dev)           # <nu, b - mu>
        pad = lambda t, val: torch.nn.functional.pad(t, (0, mp - m), value=val)      # noqa: E731
        Ec = pad(Eb, 0.0).view(-1, 128).amax(1)
        rhoc = pad(nb, 0.0).view(-1, 128).amax(1)
        # half-width terms
        h = d // 2
        Eqh, Ebh = (qt[:, :h] - qh[:, :h]).norm(dim=1), (bt[:, :h] - bh[:, :h]).norm(dim=1)
        rq, rb = qt[:, h:].norm(dim=1), bt[:, h:].norm(dim=1)
        nqh, nbh = qt[:, :h].norm(dim=1), bt[:, :h].norm(dim=1)
        Ech, Rc, rhoch = (pad(t, 0.0).view(-1, 128).amax(1) for t in (Ebh, rb, nbh))
        cand = surv = live = 0
        rows_w = 0
        for i in range(0, n, 500):
            sl = slice(i, i + 500)
            x = qh[sl] @ bh.T + beta[None, :]
            exact = qn[sl] @ bn.T
            okq = exact.amax(1) >= gate
            live += int(okq.sum())
            xc = pad(x, -9.0).view(x.shape[0], -1, 128).amax(2)                         # chunk bests
            bound = Eq[sl, None] * rhoc[None, :] + (nq[sl, None] + Eq[sl, None]) * Ec[None, :]
            up, lo = xc + bound, xc - bound
            gq = gate - alpha[sl]
            qlow = lo.amax(1)
            c = (up >= torch.maximum(qlow, gq)[:, None]) & (up.amax(1) >= gq)[:, None]
            cand += int(c.sum())
            xh = qh[sl, :h] @ bh[:, :h].T + beta[None, :]
            xhc = pad(xh, -9.0).view(x.shape[0], -1, 128).amax(2)
            bh_ = Eqh[sl, None] * rhoch[None, :] + (nqh[sl, None] + Eqh[sl, None]) * Ech[None, :] + rq[sl, None] * Rc[None, :]
            surv += int(((xhc + bh_) >= gq[:, None]).sum())
        print(f"  {vname:40s} |q~| {nq.mean():.3f} |b~| {nb.mean():.3f}  E_q {Eq.mean():.4f} E_b {Eb.mean():.4f} (max E_c {Ec.max():.4f})  "
              f"window {2 * (Eq.mean() * rhoc.mean() + (nq.mean() + Eq.mean()) * Ec.mean()):.4f}  "
              f"candidate chunks / query {cand / n:7.2f}   half-width survivors / query {surv / n:8.2f}   ({live} of {n} reach the gate)")


torch.manual_seed(0)
n, m, d = 20000, 200000, 384
for common in (1.0, 0.0):
    p = synth.make_lifted_pair_device(n, m, d, seed=42, device=dev, clouds=10, view_noise=0.1, common=common)
    evaluate(f"C2_lifted, common {common}", p["q_desc"][:NQ], p["b_desc"])
    del p
p = synth.make_pair_device(n, m, d, seed=42, device=dev)
evaluate("D.2 (the headline's data)", p["q_desc"][:NQ], p["b_desc"])
del p

# C3's data (bench.py extra_configs): the ViT's own features lifted onto 20 000 points; the map = Gaussian rows, 10 % of them the lifted rows + noise
from vfmreg import ops  # noqa: E402
from vfmreg import vit as V  # noqa: E402
rng = np.random.default_rng(0)
B, H, W = 6, 1200, 1600
imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).to(dev)
model = V.ViTS14(V.random_weights(0), H, W, device=dev)
grids = model.forward(imgs)
xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 6, n)]
pcl = torch.from_numpy(np.ascontiguousarray(np.insert(xyz, 3, 1, axis=1).T)).to(dev)
K = np.array([[800.0, 0, 800], [0, 800, 600], [0, 0, 1]])
Ps = []
for i in range(6):
    y = np.deg2rad(60 * i)
    R = np.stack([[np.sin(y), -np.cos(y), 0], [0, 0, -1], [np.cos(y), np.sin(y), 0]])
    Ps.append(K @ np.c_[R, np.zeros(3)])
desc = torch.empty((n, 384), dtype=torch.float32, device=dev)
filled = torch.zeros(n, dtype=torch.uint8, device=dev)
plan = ops.LiftPlan([dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, proj_image=None,
                          grid=grids[c], Hup=H, Wup=W, rot_mode=0, raw_image=imgs[c]) for c in range(6)], 384)
plan(pcl, desc, filled)
g = torch.Generator(device=dev).manual_seed(3)
b_desc = torch.randn(m, 384, device=dev, generator=g)
pick = torch.randperm(m, device=dev, generator=g)[:n]
b_desc[pick] = desc + 0.02 * desc.abs().mean() * torch.randn(n, 384, device=dev, generator=g)
torch.cuda.synchronize()
evaluate("C3 (ViT features lifted; map 90 % Gaussian rows)", desc[:NQ], b_desc)
# ... and a map made of lifted rows only (ten scans' worth of the same rig: what a real map of lifted features is)
bl = desc.repeat(10, 1) + 0.05 * desc.abs().mean() * torch.randn(m, 384, device=dev, generator=g)
evaluate("C3-like, every map row a lifted ViT feature", desc[:NQ], bl)
