"""The compositing scan of the AWP consumer (reference ``networks/dpnerf/awp.py``): the one piece of
``AdaptiveWeightProposal`` that is a scan over the path's per-sample outputs, forward and backward (an autograd node on
hand-written kernels).  The rest of AWP (sample / motion embedding MLPs, MotionAggregationModule) stays PyTorch in the
reference's caller."""
from __future__ import annotations

import torch

from . import _lib as L


class _FeatureIntegration(torch.autograd.Function):
    """evd_awp_feature_integration / evd_awp_feature_integration_bwd as one autograd node: gradients to the per-sample features (the
    AWP embedding MLP and, through it, the fine level), to z_vals and to rays_d (the blur kernel's ray directions)."""

    @staticmethod
    def forward(ctx, f, z, d):
        N, S, Cc = f.shape
        out = torch.empty((N, Cc), dtype=torch.float32, device=f.device)
        L.check(L.lib().evd_awp_feature_integration(L.ptr(f), L.ptr(z), L.ptr(d), N, S, Cc, L.ptr(out), L.stream_ptr()),
                "evd_awp_feature_integration")
        ctx.save_for_backward(f, z, d)
        return out

    @staticmethod
    def backward(ctx, g):
        f, z, d = ctx.saved_tensors
        N, S, Cc = f.shape
        g = g.contiguous().float()
        df = torch.empty_like(f)
        dz = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        dd = torch.empty_like(d) if ctx.needs_input_grad[2] else None
        L.check(L.lib().evd_awp_feature_integration_bwd(L.ptr(f), L.ptr(z), L.ptr(d), L.ptr(g), N, S, Cc, L.ptr(df), L.ptr(dz), L.ptr(dd),
                                                        L.stream_ptr()), "evd_awp_feature_integration_bwd")
        return df, dz, dd


def feature_integration(feat, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """awp.py:49-77: feat [N_rays, N_motion, N_samples, C], z_vals [N_rays*N_motion, N_samples], rays_d [N_rays*N_motion, 3]
    -> [N_rays, N_motion, C].  Restated as written in the reference: every channel is its own density, the last alpha is 0,
    and the cumprod of awp.py:69-73 runs along the channel axis (dim -1) of the previous sample's row."""
    n_rays, n_motion, S, Cc = feat.shape
    f = feat.reshape(-1, S, Cc).contiguous().float()
    z = z_vals.reshape(-1, S).contiguous().float()
    d = rays_d.reshape(-1, 3).contiguous().float()
    N = f.shape[0]
    if z.shape[0] != N or d.shape[0] != N:
        raise L.EvdError("feature_integration: z_vals / rays_d need one row per (ray, motion)")
    return _FeatureIntegration.apply(f, z, d).reshape(n_rays, n_motion, Cc)
