#!/usr/bin/env python3
"""debug: the small fp32 training operators of the FC encoders at row counts around the EgoCap batch sizes, against float64"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from egotap_amd import train_ops as T
from gpu_util import lift_net
net, _, _ = lift_net("UnrealEgo")
h = net._ensure_handle()
torch.manual_seed(1)
def rel(a, b):
    return float((a.double().cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-30))
for M in (60, 68, 90, 96, 97, 102, 128, 129, 136, 200):
    row = [f"M={M}"]
    for N in (128, 512, 2048):
        z = torch.randn(M, N, device="cuda")
        g = torch.rand(N, device="cuda") + 0.5
        bta = torch.randn(N, device="cuda")
        y, mean, rstd = T.bn_lrelu_fwd(z.clone(), g, bta, torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"))
        dy = torch.randn(M, N, device="cuda") + 0.3
        dg, db = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        dz = T.bn_lrelu_bwd(z, y, dy, g, mean, rstd, dg, db)
        zd = z.double().cpu().requires_grad_(True)
        mu = zd.mean(0); var = ((zd - mu) ** 2).mean(0)
        gd, bd = g.double().cpu().requires_grad_(True), bta.double().cpu().requires_grad_(True)
        yd = torch.nn.functional.leaky_relu((zd - mu) / torch.sqrt(var + 1e-5) * gd + bd, 0.2)
        dzr, dgr, dbr = torch.autograd.grad(yd, (zd, gd, bd), dy.double().cpu())
        row.append(f"bn{N}: dz {rel(dz, dzr):.1e} dg {rel(dg, dgr):.1e} db {rel(db, dbr):.1e}")
        cs = torch.empty(N, device="cuda")
        T.colsum(dz, cs, M, N)
        row.append(f"cs {rel(cs, dz.double().cpu().sum(0)):.1e}")
    for (N, K) in ((512, 128), (2048, 512)):
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
        out = T.gemm_nt(h, a, w, None, M, N, K, epi=T.TE_NONE)
        row.append(f"nt{N}x{K} {rel(out, a.double().cpu() @ w.double().cpu().T):.1e}")
        dyy = torch.randn(M, N, device="cuda")
        dw = torch.empty(N, K, device="cuda")
        T.gemm_tn(h, dyy, a, dw, M, N, K)
        row.append(f"tn{N}x{K} {rel(dw, dyy.double().cpu().T @ a.double().cpu()):.1e}")
    print("  ".join(row), flush=True)
