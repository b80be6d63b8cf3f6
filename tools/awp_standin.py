"""A PyTorch module with the attribute surface and call contract of the reference's AdaptiveWeightProposal (networks/dpnerf/awp.py:9-47,
79-117) for the GPU box, where the reference itself cannot be imported: the per-sample embedding is the reference's structure exactly
(sample_feature_embed_layer: 4 x Linear + ReLU, 128 -> 64); the motion aggregation module is a small stand-in with the MAM's call
contract (x_global [R, P, C], x_local [R P, S, 64]) -> [R, P, C].  Used by tests/test_gpu_awp.py and tools/bench_train_step.py --awp
(wrapped in evdeblurnerf_amd.awp.FusedAWP, or called directly as the plain-torch comparison)."""
import torch


def scan_as_written(f, z, d):
    """awp.py:58-75 AS WRITTEN (zeros appended, cumprod along the channel axis of the previous sample's row)"""
    dists = (z[..., 1:] - z[..., :-1]) * torch.norm(d[..., None, :], dim=-1)
    alpha = -torch.exp(-f[..., :-1, :] * dists[..., None]) + 1
    alpha = torch.cat([alpha, torch.zeros_like(alpha[:, 0:1])], dim=-2)
    wts = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1, alpha.shape[-1]), dtype=f.dtype, device=f.device), -alpha + (1. + 1e-10)], -2), -1)[:, :-1, :]
    return torch.sum(wts * f, dim=-2)


class CorrLike(torch.nn.Module):
    """Plain-torch restatement of the reference's CorrelationModule (mam.py:13-53) with its parameter names, so that the reference's
    state dict loads (tests/test_oracle_golden.py pins this module to the golden G22 the real module produced)."""

    def __init__(self, ch):
        super().__init__()
        nn, mid = torch.nn, ch // 2
        self.conva, self.convb, self.convc = (nn.Conv1d(ch, mid, 1, bias=False) for _ in range(3))
        self.convn, self.convl = nn.Conv1d(mid, mid, 1, bias=False), nn.Conv1d(mid, mid, 1, bias=False)
        self.convd = nn.Sequential(nn.Conv1d(2 * mid, ch, 1, bias=False), nn.BatchNorm1d(ch))
        self.line_conv_att = nn.Conv2d(ch, 1, 1, bias=False)

    def forward(self, x, curves):                  # x [B, C, P], curves [B, C, P, S]
        att = self.line_conv_att(curves)
        k_p = self.conva((curves * att.softmax(-1)).sum(-1))           # weights along the samples -> [B, mid, P]
        k_s = self.convb((curves * att.softmax(-2)).sum(-2))           # weights along the sub-exposures -> [B, mid, S]
        q = self.convc(x).transpose(1, 2)
        f_p = (q @ k_p).softmax(-1) @ self.convn(k_p).transpose(1, 2)
        f_s = (q @ k_s).softmax(-1) @ self.convl(k_s).transpose(1, 2)
        return torch.nn.functional.leaky_relu(x + self.convd(torch.cat([f_p, f_s], -1).transpose(1, 2)), 0.2)


class MAMLike(torch.nn.Module):
    """... and of MotionAggregationModule (mam.py:56-83): x_global [R, P, C], x_local [R P, S, 64] -> [R, P, C]"""

    def __init__(self, ch=32, num_motion=4):
        super().__init__()
        nn = torch.nn
        self.Corr, self.linear, self.num_motion = CorrLike(ch), nn.Linear(64, 32), num_motion
        self.conv = nn.Sequential(nn.Conv2d(2 * ch, ch, 1, bias=False), nn.BatchNorm2d(ch), nn.LeakyReLU(0.2))    # present, unused (as there)

    def forward(self, x_global, x_local):
        P = self.num_motion + 1
        loc = self.linear(x_local.reshape(-1, P, *x_local.shape[1:]))                  # [R, P, S, 32]
        return self.Corr(x_global.transpose(1, 2), loc.permute(0, 3, 1, 2)).transpose(1, 2)


class RefLikeAWP(torch.nn.Module):
    """A module with the reference AdaptiveWeightProposal's attribute surface (awp.py:9-47) for FusedAWP to wrap on the GPU box (the
    reference itself cannot travel there): the per-sample embedding is the reference's structure exactly; the motion aggregation
    module is MAMLike (mam="corr", the reference's structure) or a small stand-in with only the MAM's call contract
    (x_global [R, P, C], x_local [R P, S, 64]) -> [R, P, C] (mam="mean")."""

    def __init__(self, P=5, W_mot=32, view_ch=4, mam="mean"):
        super().__init__()
        self.output_ch, self.ccw_fine_scale, self.ray_dir_freq = P, 0.05, 2
        ch = 3 * (1 + 2 * 2)            # a differentiable 2-frequency encoding (the reference's get_embedder(ray_dir_freq) is torch too)
        self.ray_dirs_embed_fn = lambda x: torch.cat([x] + [f(x * 2.0 ** k) for k in range(2) for f in (torch.sin, torch.cos)], -1)
        self.sample_feature_embed_layer = torch.nn.ModuleList([torch.nn.Linear(128, 64)] + [torch.nn.Linear(64, 64) for _ in range(3)])
        self.motion_feature_embed_layer = torch.nn.ModuleList([torch.nn.Linear(64 + view_ch + ch, W_mot), torch.nn.Linear(W_mot, W_mot)])
        self.w_linear = torch.nn.Linear(W_mot, P)
        if mam == "corr":                # the reference's MotionAggregationModule structure (FusedAWP runs its per-sample part on the library)
            self.MAM = MAMLike(W_mot, P - 1)
        else:                            # a MAM FusedAWP does not know: it is called as it is
            self.local = torch.nn.Linear(64, W_mot)
            self.MAM = self._mean_mam

    def _mean_mam(self, x_global, x_local):
        loc = self.local(x_local).mean(1).reshape(x_global.shape)
        return torch.nn.functional.leaky_relu(x_global + loc, 0.2)

    def forward(self, depth_feature, z_vals, rays_d, view_feature):         # awp.py:79-117 in plain torch (float32 reference of the test)
        P = self.output_ch
        h = depth_feature
        for l in self.sample_feature_embed_layer:
            h = torch.relu(l(h))
        return self.forward_from_local(h, z_vals, rays_d, view_feature)

    def forward_from_local(self, h_local, z_vals, rays_d, view_feature):    # awp.py:89-95, 102-117 behind the per-sample embedding
        P = self.output_ch
        n_ray = h_local.shape[0] // P
        dirs = rays_d.reshape(n_ray, P, -1)[:, 0, :]
        enc = self.ray_dirs_embed_fn(dirs / torch.norm(dirs, dim=-1, keepdim=True))
        view = enc if view_feature is None else torch.cat([view_feature, enc], -1)
        h = scan_as_written(h_local, z_vals, rays_d).reshape(n_ray, P, -1)
        h = torch.cat([h, view.unsqueeze(1).repeat(1, P, 1)], -1)
        for l in self.motion_feature_embed_layer:
            h = torch.relu(l(h))
        h = self.MAM(h, h_local)
        w = torch.sigmoid(self.w_linear(h.mean(1)))
        return w / w.sum(-1, keepdim=True)
