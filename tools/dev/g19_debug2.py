import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from types import SimpleNamespace
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.renderer import NeRFAll
import evdeblurnerf_amd.voxnerf as V
from test_gpu_train import AABB
from conftest import load_golden
g = load_golden("G19_c2f_grads")
gc, gf = [int(v) for v in g["grid_coarse"]], [int(v) for v in g["grid_fine"]]
sd = dict(W.prefixed(W.make_pdrf_state_dict(91, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15, add_bias_color=True), "mlp_coarse"))
sd.update(W.prefixed(W.make_pdrf_state_dict(92, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True), "mlp_fine"))
args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=16, kernel_type="RBK", kernel_use_awp=False,
                       rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                       coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=24 ** 3,
                       kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                       fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=48 ** 3)
captured = {}
orig = V._VoxelMLP.backward
def spy(ctx, d_raw, d_feature=None):
    out = orig(ctx, d_raw, d_feature)
    net = ctx.net
    rec = {"d_raw": d_raw.detach().clone(), "d_fts": out[1].detach().clone(), "d_pts": out[2].detach().clone() if out[2] is not None else None}
    if out[0] is not None:
        for k, v in net.unflatten(out[0].detach().clone()).items():
            rec["w." + k] = v
    captured.setdefault(captured["tag"], []).append(rec)
    return out
V._VoxelMLP.backward = staticmethod(spy)
def run(prec, R):
    rays_np = g["rays"] if R == 24 else W.synthetic_rays(19, R)
    rs = np.random.RandomState(1901)
    w_rgb, w_rgb0 = rs.standard_normal((R, 3)).astype(np.float32), rs.standard_normal((R, 3)).astype(np.float32)
    model = NeRFAll(args, sd, precision=prec).enable_training(sd).train()
    rays = torch.tensor(rays_np, device="cuda", requires_grad=True)
    captured["tag"] = (prec, R)
    rgb, rgb0, other, _ = model(400, 400, W.synthetic_camera(), 1 << 20, rays=rays, ndc=True, near=0., far=1., N_samples=16, N_importance=16, perturb=0., raw_noise_std=0.)
    loss = (rgb * torch.tensor(w_rgb, device="cuda")).sum() + (rgb0 * torch.tensor(w_rgb0, device="cuda")).sum() + 0.1 * other["TV"].sum()
    loss.backward()
    out = {"rays": rays.grad.detach().clone()}
    for k, v in model.named_parameters():
        out[k] = v.grad.detach().clone()
    return out
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for R in (2048,):
    ref = run("f16x3", R)
    for prec in ("f16c", "f16"):
        got = run(prec, R)
        errs = {k: rel(got[k], ref[k]) for k in ref}
        print(f"R={R} {prec} vs f16x3 (element-wise relative L2):", {k: f"{v:.1e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:12]})
        for i, (a, b) in enumerate(zip(captured[(prec, R)], captured[("f16x3", R)])):
            print(f"   level call {i}: shape {tuple(b['d_raw'].shape)}:", {k: f"{rel(a[k], b[k]):.1e}" for k in b if b[k] is not None})
            nz = (b["d_raw"].abs().amax(-1) > 1e-6 * b["d_raw"].abs().max()).sum().item()
            print(f"      samples with |d_raw| > 1e-6 max: {nz} of {b['d_raw'].shape[0] * b['d_raw'].shape[1]}")
