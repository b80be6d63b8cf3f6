"""float64 torch restatements of the differentiable pieces of the path, written from the reference's formulas (file:line in
each docstring).  They are the autograd comparator of the GPU training tests (tests/test_gpu_train.py); they are themselves
pinned to gradients computed by the REAL reference (tests/golden/G18, G19; tests/test_oracle_golden.py) -- the chain is
reference --(1e-4)--> restatement --(1e-3, same ReLU pattern)--> HIP kernels.  Test infrastructure only."""
import numpy as np
import torch
import torch.nn.functional as Fn


def embed(x, L):
    """Embedder.forward, networks/embedding.py:88-98: [x, sin(x 2^k), cos(x 2^k), ...]"""
    out = [x]
    for k in range(L):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


class TorchNerf(torch.nn.Module):
    """NeRF.mlpforward + eval, networks/nerf.py:46-72,131-162 (skips = [4], use_viewdirs); any width"""

    def __init__(self, sd):
        super().__init__()
        self.p = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(torch.tensor(np.asarray(v), dtype=torch.float64))
                                         for k, v in sd.items()})
        self.D = sum(1 for k in sd if k.startswith("pts_linears.") and k.endswith(".weight"))

    def lin(self, name, x):
        y = x @ self.p[name + "_weight"].T
        return y + self.p[name + "_bias"] if name + "_bias" in self.p else y

    def forward(self, pts, dirs, keep=None, masks=None):
        """masks: {name: 0/1 tensor} replaces the ReLU of that layer by a multiplication (same derivative pattern as the kernel's)"""
        pe, ped = embed(pts, 10), embed(dirs, 4)
        h = pe
        for l in range(self.D):
            h = self.lin(f"pts_linears_{l}", h)
            h = torch.relu(h) if masks is None else h * masks[f"h{l}"]
            if keep is not None:
                keep[f"h{l}"] = h
            if l == 4:
                h = torch.cat([pe, h], -1)
        alpha = self.lin("alpha_linear", h)
        f = self.lin("feature_linear", h)
        hv = self.lin("views_linears_0", torch.cat([f, ped], -1))
        hv = torch.relu(hv) if masks is None else hv * masks["hv"]
        if keep is not None:
            keep["f"], keep["hv"] = f, hv
        return torch.cat([self.lin("rgb_linear", hv), alpha], -1)


def nerf_composite(raw, z, rays_d):
    """NeRF.raw2outputs, networks/nerf.py:74-129 (sigmoid colour, relu density, no noise / white background) -> rgb_map, weights"""
    dists = (z[:, 1:] - z[:, :-1]) * rays_d.norm(dim=-1, keepdim=True)
    alpha = torch.cat([1 - torch.exp(-torch.relu(raw[:, :-1, 3]) * dists), torch.ones_like(dists[:, :1])], -1)
    w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    return (w[..., None] * torch.sigmoid(raw[..., :3])).sum(-2), w


class TorchVoxLevel(torch.nn.Module):
    """the per-sample part of VoxelNeRFBase.forward, networks/pdrf/voxnerf.py:210-221,240-254"""
    KEYS = ["sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.0.bias", "color_net.1.weight",
            "color_net.1.bias", "color_net.2.weight", "color_net.2.bias"]

    def __init__(self, sd):
        super().__init__()
        self.p = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(torch.tensor(np.asarray(sd[k]), dtype=torch.float64))
                                         for k in self.KEYS if k in sd})

    def cl(self, i, x):
        y = x @ self.p[f"color_net_{i}_weight"].T
        return y + self.p[f"color_net_{i}_bias"] if f"color_net_{i}_bias" in self.p else y

    def forward(self, pts, dirs, fts, masks=None, want_geo=False, keep=None):
        """masks: 0/1 ReLU patterns to use instead of the own ones; keep: dict that receives the pre-activations of hid / c0 / c1"""
        P = self.p

        def act(x, k):
            if keep is not None:
                keep[k] = x.detach()
            return torch.relu(x) if masks is None else x * masks[k]

        h = act(torch.cat([fts, embed(pts, 10)], -1) @ P["sigma_net_0_weight"].T, "hid")
        sg = h @ P["sigma_net_1_weight"].T
        c = act(self.cl(0, torch.cat([sg[:, 1:], embed(dirs, 4)], -1)), "c0")
        c = act(self.cl(1, c), "c1")
        raw = torch.cat([sg[:, :1], torch.sigmoid(self.cl(2, c))], -1)
        return (raw, sg[:, 1:]) if want_geo else raw


def torch_appfeature(planes, lines, basis, pts, aabb):
    """VoxelNeRFBase.sample / compute_appfeature, voxnerf.py:132-151,203-208; reference layouts ([1,C,H,W] planes, [1,C,L,1] lines)"""
    lo, hi = torch.tensor(aabb[0], dtype=torch.float64), torch.tensor(aabb[1], dtype=torch.float64)
    xyz = (pts - lo) * (2.0 / (hi - lo)) - 1
    mat, vec = [[0, 1], [0, 2], [1, 2]], [2, 1, 0]
    pc, lc = [], []
    for i in range(3):
        cp = xyz[:, mat[i]].view(1, -1, 1, 2)
        cl = torch.stack([torch.zeros_like(xyz[:, vec[i]]), xyz[:, vec[i]]], -1).view(1, -1, 1, 2)
        pc.append(Fn.grid_sample(planes[i], cp, align_corners=True).view(-1, pts.shape[0]))
        lc.append(Fn.grid_sample(lines[i], cl, align_corners=True).view(-1, pts.shape[0]))
    return (torch.cat(pc) * torch.cat(lc)).T @ basis.T


def torch_tv(x):
    """TVLoss.forward, voxnerf.py:306-324 (weight 1, batch 1)"""
    ch = x.shape[1] * (x.shape[2] - 1) * x.shape[3]
    cw = max(x.shape[1] * x.shape[2] * (x.shape[3] - 1), 1)
    return 2 * (((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum() / ch + ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum() / cw)


def vox_composite(raw, z, rays_d):
    """VoxelNeRFBase.raw2outputs, voxnerf.py:153-201 (sigma channel 0, colours 1:, relu density, training) -> rgb_map, weights"""
    dists = (z[:, 1:] - z[:, :-1]) * rays_d.norm(dim=-1, keepdim=True)
    dens = torch.relu(raw[:, :-1, 0])
    alpha = torch.cat([1 - torch.exp(-dens * dists), torch.ones_like(dens[:, :1])], -1)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    return (w[..., None] * raw[..., 1:]).sum(-2), w


def c2f_pipeline(levels, grids, rb, z0, zm, aabb):
    """render_rays, mode='c2f', networks/renderer.py:182-217, on GIVEN sample positions (z0 coarse, zm merged; the reference
    detaches the resampled positions).  levels = {"coarse"|"fine": TorchVoxLevel}, grids = {name: (planes, lines, basis)},
    rb [R,11] ray batch.  -> (rgb0, rgb)"""
    R, S, St = rb.shape[0], z0.shape[1], zm.shape[1]
    o, d, vd = rb[:, None, 0:3], rb[:, None, 3:6], rb[:, 8:11]

    def feat(name, pts):
        pl, li, ba = grids[name]
        return torch_appfeature(pl, li, ba, pts.reshape(-1, 3), aabb)

    pts0 = o + d * z0[..., None]
    raw0 = levels["coarse"](pts0.reshape(-1, 3), vd[:, None].expand(-1, S, -1).reshape(-1, 3), feat("coarse", pts0)).reshape(R, S, 4)
    rgb0, _ = vox_composite(raw0, z0, d[:, 0])
    ptm = o + d * zm[..., None]
    ftm = torch.cat([feat("coarse", ptm), feat("fine", ptm)], -1)
    raw1 = levels["fine"](ptm.reshape(-1, 3), vd[:, None].expand(-1, St, -1).reshape(-1, 3), ftm).reshape(R, St, 4)
    rgb1, _ = vox_composite(raw1, zm, d[:, 0])
    return rgb0, rgb1


def grad_elements(g, seed, k=256):
    """element-level pins of a gradient tensor: (flat indices, values) of its k largest-magnitude elements and of k seeded random ones"""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    k = min(k, g.size)
    top = np.argsort(-np.abs(g), kind="stable")[:k]
    rnd = np.random.RandomState(seed).choice(g.size, size=k, replace=g.size < k)
    idx = np.concatenate([top, rnd]).astype(np.int64)
    return idx, g[idx].copy()


def check_grad_elements(got, idx, val, tol):
    """max |got[idx] - val| relative to the largest pinned element"""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    scale = max(float(np.abs(val).max()), 1e-30)
    err = float(np.abs(got[idx] - val).max()) / scale
    return err, err < tol


def grad_summary(g, seed):
    """(L2 norm, projection on a seeded random direction, first 32 elements) of a gradient tensor: what the goldens store"""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    v = np.random.RandomState(seed).standard_normal(g.size)
    return np.array([np.sqrt((g * g).sum()), (g * v).sum()]), g[:32].copy()
