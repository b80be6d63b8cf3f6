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


class RefLikeAWP(torch.nn.Module):
    """A module with the reference AdaptiveWeightProposal's attribute surface (awp.py:9-47) for FusedAWP to wrap on the GPU box (the
    reference itself cannot travel there): the per-sample embedding is the reference's structure exactly; the motion aggregation
    module is a small stand-in with the MAM's call contract (x_global [R, P, C], x_local [R P, S, 64]) -> [R, P, C]."""

    def __init__(self, P=5, W_mot=32, view_ch=4):
        super().__init__()
        self.output_ch, self.ccw_fine_scale = P, 0.05
        ch = 3 * (1 + 2 * 2)            # a differentiable 2-frequency encoding (the reference's get_embedder(ray_dir_freq) is torch too)
        self.ray_dirs_embed_fn = lambda x: torch.cat([x] + [f(x * 2.0 ** k) for k in range(2) for f in (torch.sin, torch.cos)], -1)
        self.sample_feature_embed_layer = torch.nn.ModuleList([torch.nn.Linear(128, 64)] + [torch.nn.Linear(64, 64) for _ in range(3)])
        self.motion_feature_embed_layer = torch.nn.ModuleList([torch.nn.Linear(64 + view_ch + ch, W_mot), torch.nn.Linear(W_mot, W_mot)])
        self.local = torch.nn.Linear(64, W_mot)
        self.w_linear = torch.nn.Linear(W_mot, P)

    def MAM(self, x_global, x_local):
        loc = self.local(x_local).mean(1).reshape(x_global.shape)
        return torch.nn.functional.leaky_relu(x_global + loc, 0.2)

    def forward(self, depth_feature, z_vals, rays_d, view_feature):         # awp.py:79-117 in plain torch (float32 reference of the test)
        P = self.output_ch
        n_ray = depth_feature.shape[0] // P
        dirs = rays_d.reshape(n_ray, P, -1)[:, 0, :]
        view = torch.cat([view_feature, self.ray_dirs_embed_fn(dirs / torch.norm(dirs, dim=-1, keepdim=True))], -1)
        h = depth_feature
        for l in self.sample_feature_embed_layer:
            h = torch.relu(l(h))
        h_local = h
        h = scan_as_written(h, z_vals, rays_d).reshape(n_ray, P, -1)
        h = torch.cat([h, view.unsqueeze(1).repeat(1, P, 1)], -1)
        for l in self.motion_feature_embed_layer:
            h = torch.relu(l(h))
        h = self.MAM(h, h_local)
        w = torch.sigmoid(self.w_linear(h.mean(1)))
        return w / w.sum(-1, keepdim=True)
