#!/usr/bin/env python
"""CPU emulation of the MLP arithmetic modes on the 8x256 NeRF: where does the half-precision RGB error come from, and what does a
low-precision (fp6 e2m3, block-scaled) correction product buy?  float64 torch is the reference.  Used to choose the `f16c` mode.
    python tools/precision_anatomy.py [--scale 1.4] [--weights file.npz] [--rays 256]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from evdeblurnerf_amd import weights as W

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--weights", default=None)
ap.add_argument("--rays", type=int, default=256)
ap.add_argument("--samples", type=int, default=128)
a = ap.parse_args()
torch.set_num_threads(8)
F64 = torch.float64

def h(x):  # round to f16 (saturating), back to f64
    return x.to(torch.float32).clamp(-65504, 65504).to(torch.float16).to(F64)

E2M3 = torch.tensor(sorted({(m / 8.0) if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1) for e in range(4) for m in range(8)}), dtype=F64)
E3M2 = torch.tensor(sorted({(m / 4.0) * 0.25 if e == 0 else (1 + m / 4.0) * 2.0 ** (e - 3) for e in range(8) for m in range(4)}), dtype=F64)
E2M1 = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6], dtype=F64)

def q_block(x, grid, block=32, dim=-1):
    """block-scaled quantisation along `dim`: power-of-two scale per `block` elements so that the block max fits the grid"""
    x = x.transpose(dim, -1)
    shp = x.shape
    K = shp[-1]
    pad = (-K) % block
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp(min=1e-300)
    gmax = float(grid[-1])
    e = torch.ceil(torch.log2(amax / gmax))
    sc = 2.0 ** e
    v = (xb / sc).abs().clamp(max=gmax)
    idx = torch.searchsorted(grid, v.contiguous()).clamp(1, len(grid) - 1)
    lo, hi = grid[idx - 1], grid[idx]
    qv = torch.where((v - lo) > (hi - v), hi, lo) * torch.sign(xb) * sc
    qv = qv.reshape(*x.shape)[..., :K].reshape(shp)
    return qv.transpose(dim, -1)

class Mode:
    """y = W x with: main product f16(W) f16(x) [or exact], + optional corrections"""
    def __init__(s, name, w="h", x="h", cw=None, cx=None, grid=E2M3, layers=None, wblock=32, rtz=False, wgrid=None):
        s.name, s.w, s.x, s.cw, s.cx, s.grid, s.layers, s.wblock, s.rtz, s.wgrid = name, w, x, cw, cx, grid, layers, wblock, rtz, (wgrid if wgrid is not None else grid)
    def lin(s, Wt, b, x, li):
        full = s.layers is None or li in s.layers
        Wh = h(Wt) if s.w == "h" else Wt
        xh = h(x) if s.x == "h" else x
        y = xh @ Wh.T + b
        if full and s.cw:     # correction for the weight rounding: (W - Wh) x
            Wl = Wt - Wh
            if s.cw == "exact": y = y + xh @ Wl.T
            else: y = y + q_block(xh, s.grid) @ q_block(Wl, s.wgrid, block=s.wblock or Wl.shape[-1]).T
        if full and s.cx:     # correction for the activation rounding: W (x - xh)
            xl = x - xh
            if s.cx == "exact": y = y + xl @ Wh.T
            else: y = y + q_block(xl, s.grid) @ q_block(Wh, s.wgrid, block=s.wblock or Wh.shape[-1]).T
        return y

def embed(x, L):
    out = [x]
    for i in range(L):
        out += [torch.sin(x * 2.0 ** i), torch.cos(x * 2.0 ** i)]
    return torch.cat(out, -1)

def render(sd, rb, z, mode):
    g = lambda k: torch.as_tensor(sd["mlp_coarse." + k], dtype=F64)
    R, S = z.shape
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    pe = embed(pts.reshape(-1, 3), 10)
    ve = embed(rb[:, 8:11], 4)[:, None, :].expand(R, S, 27).reshape(-1, 27)
    x = pe
    for i in range(8):
        x = torch.relu(mode.lin(g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias"), x, i))
        if i == 4:
            x = torch.cat([pe, x], -1)
    alpha = mode.lin(g("alpha_linear.weight"), g("alpha_linear.bias"), x, 8)
    feat = mode.lin(g("feature_linear.weight"), g("feature_linear.bias"), x, 9)
    hv = torch.relu(mode.lin(g("views_linears.0.weight"), g("views_linears.0.bias"), torch.cat([feat, ve], -1), 10))
    rgb = mode.lin(g("rgb_linear.weight"), g("rgb_linear.bias"), hv, 11)
    raw = torch.cat([rgb, alpha], -1).reshape(R, S, 4)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, dtype=F64)], -1) * rb[:, None, 3:6].norm(dim=-1)
    al = 1 - torch.exp(-torch.relu(raw[..., 3]) * dists)
    T = torch.cumprod(torch.cat([torch.ones((R, 1), dtype=F64), 1 - al + 1e-10], -1), -1)[:, :-1]
    w = al * T
    return (w[..., None] * torch.sigmoid(raw[..., :3])).sum(1), raw

if a.weights:
    sd = dict(np.load(a.weights))
else:
    sd = W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
    for i in range(8):
        sd[f"mlp_coarse.pts_linears.{i}.weight"] = sd[f"mlp_coarse.pts_linears.{i}.weight"] * a.scale
from evdeblurnerf_amd.renderer import NeRFAll
K = W.synthetic_camera()
rays = torch.as_tensor(W.synthetic_rays(77, a.rays))
rb = NeRFAll.ray_batch_train(400, 400, K, rays).to(F64)
z = torch.linspace(0, 1, a.samples, dtype=F64).expand(a.rays, a.samples).contiguous()
z = rb[:, 6:7] * (1 - z) + rb[:, 7:8] * z
ref, raw_ref = render(sd, rb, z, Mode("exact", w="e", x="e"))
print(f"sigma max {float(raw_ref[..., 3].max()):.2f}  rgb range {float(ref.min()):.3f}..{float(ref.max()):.3f}")
modes = [Mode("f16 (Wh Xh)"),
         Mode("exact W, f16 x", w="e"),
         Mode("f16 W, exact x", x="e"),
         Mode("f16 + Wl.Xh exact", cw="exact"),
         Mode("f16 + Wh.Xl exact", cx="exact"),
         Mode("f16 + both exact (f16x3)", cw="exact", cx="exact"),
         Mode("f16 + fp6 e2m3 both", cw="q", cx="q"),
         Mode("f16 + fp6 e3m2 both", cw="q", cx="q", grid=E3M2),
         Mode("f16 + fp4 e2m1 both", cw="q", cx="q", grid=E2M1),
         Mode("fp6 e2m3 both, W scale per row", cw="q", cx="q", wblock=0),
         Mode("W operands fp4 per row, X fp6", cw="q", cx="q", wblock=0, wgrid=E2M1),
         Mode("f16 + fp6 e2m3 Wh.Xl only", cx="q"),
         Mode("f16 + fp6 e2m3 Wl.Xh only", cw="q"),
         ]
for m in modes:
    rgb, raw = render(sd, rb, z, m)
    print(f"{m.name:34s} rgb Linf {float((rgb - ref).abs().max()):.3e}   raw Linf {float((raw - raw_ref).abs().max()):.3e}")
