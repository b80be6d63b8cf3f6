"""Forward and gradient parity of the training modes on a 65 536-sample c2f batch (2048 rays x (16 + 16) samples, small grids), against the
float32-grade mode f16x3 -- which equals torch.autograd ON THE REFERENCE to 2e-5 of the gradient norm on golden G19
(tests/test_gpu_train_f32grade.py).  Used by bench.py's train_iteration leg and tests/test_gpu_train_f16c.py.  GPU only."""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def c2f_gradient_parity(precs=("f16c", "f16m", "f16"), R=2048, reference="f16x3"):
    """-> {prec: {"rgb_linf": max |rgb - rgb_ref|, "grad_rel_l2_worst": worst relative L2 error of a gradient tensor (30 parameter tensors
    + the rays), "worst_tensor": its name}}"""
    from evdeblurnerf_amd.renderer import NeRFAll
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], 24 ** 3), W.pdrf_grid_size(AABB[0], AABB[1], 48 ** 3)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(91, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15, add_bias_color=True), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(92, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True), "mlp_fine"))
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=16, kernel_type="RBK", kernel_use_awp=False,
                           rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                           coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=24 ** 3,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                           fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=48 ** 3)
    rs = np.random.RandomState(1901)
    w_rgb, w_rgb0 = rs.standard_normal((R, 3)).astype(np.float32), rs.standard_normal((R, 3)).astype(np.float32)

    def run(p):
        model = NeRFAll(args, sd, precision=p).enable_training(sd).train()
        rays = torch.tensor(W.synthetic_rays(19, R), device="cuda", requires_grad=True)
        rgb, rgb0, other, _ = model(400, 400, W.synthetic_camera(), 1 << 20, rays=rays, ndc=True, near=0., far=1., N_samples=16, N_importance=16,
                                    perturb=0., raw_noise_std=0.)
        ((rgb * torch.tensor(w_rgb, device="cuda")).sum() + (rgb0 * torch.tensor(w_rgb0, device="cuda")).sum() + 0.1 * other["TV"].sum()).backward()
        out = {"rays": rays.grad.detach().clone()}
        out.update({k: v.grad.detach().clone() for k, v in model.named_parameters()})
        return out, rgb.detach()
    ref, rgb_ref = run(reference)
    res = {}
    for p in precs:
        got, rgb = run(p)
        errs = {k: float((got[k].double() - ref[k].double()).norm() / ref[k].double().norm()) for k in ref}
        worst = max(errs, key=errs.get)
        res[p] = {"rgb_linf": float((rgb - rgb_ref).abs().max()), "grad_rel_l2_worst": errs[worst], "worst_tensor": worst,
                  "grad_rel_l2_median": float(np.median(list(errs.values())))}
    return res


if __name__ == "__main__":
    for k, v in c2f_gradient_parity().items():
        print(k, v)
