"""Gradients of a small c2f model and of the AWP embedding through whichever backward forms the environment selects
(EVD_BWD_FUSE64, EVD_BWD_FUSE_SG, EVD_BWD_YGEN, EVD_AWP_BWD_FUSE; default: the round-5 fused forms) -> an .npz of every gradient tensor.
tests/test_gpu_bwd_fusion.py runs it twice (all fused / all per-layer) and compares the files: the fused kernels do the SAME float16
arithmetic as the chains they replace, only the summation order of the weight-gradient partials differs.
    python tools/dev/bwd_fusion_ab.py out.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402


def c2f_grads(prec, out):
    import test_gpu_train as T
    model, sd = T._c2f_model(prec, 32, 24 ** 3, 48 ** 3)
    model.train()
    pc, pf = model.trainable_parameters(sd)
    # 2501 x 24 = 60 024 coarse samples = 1876 tiles (ragged last one): more tiles than the fused kernel has wavefronts (1024), so its
    # persistent loop with the loads of the NEXT tile in flight runs; small: one tile per wavefront at most
    R, S, Ni = (333 if os.environ.get("AB_SMALL") else 2501), 24, 16
    rb = torch.tensor(T._c2f_rays(R, 4), device="cuda", requires_grad=True)
    tgt = torch.tensor(np.random.RandomState(5).uniform(0, 1, (R, 3)).astype(np.float32), device="cuda")
    res = model.render_rays_train(rb, pc, pf, S, Ni)
    loss = ((res["rgb_map"] - tgt) ** 2).mean() + ((res["rgb0"] - tgt) ** 2).mean()
    loss.backward()
    out[f"{prec}.rays"] = rb.grad.cpu().numpy()
    for name, lvl in (("coarse", pc), ("fine", pf)):
        out[f"{prec}.{name}.net"] = lvl["net"].grad.cpu().numpy()
        for i, g in enumerate(lvl["grids"]):
            out[f"{prec}.{name}.grid{i}"] = g.grad.cpu().numpy()


def awp_grads(prec, out):
    from evdeblurnerf_amd.awp import SampleFeatureEmbed
    sd = W.make_awp_embed_state_dict(211)
    ws = [sd[f"sample_feature_embed_layer.{l}.weight"] for l in range(4)]
    bs = [sd[f"sample_feature_embed_layer.{l}.bias"] for l in range(4)]
    emb = SampleFeatureEmbed(ws, bs, precision=prec)
    flat = torch.cat([torch.tensor(t).reshape(-1) for l in range(4) for t in (ws[l], bs[l])]).cuda().requires_grad_(True)
    rs = np.random.RandomState(7)
    n = (40 if os.environ.get("AB_SMALL") else 1200) * 128 - 19      # ragged last tile; 4800 tiles: several per wavefront
    x = torch.tensor((rs.standard_normal((n, 128)) * 0.7).astype(np.float32), device="cuda", requires_grad=True)
    w = torch.tensor((rs.standard_normal((n, 64)) * 1e-3).astype(np.float32), device="cuda")
    h = emb(flat, x)
    (h * w).sum().backward()
    out[f"awp.{prec}.params"] = flat.grad.cpu().numpy()
    out[f"awp.{prec}.x"] = x.grad.cpu().numpy()


if __name__ == "__main__":
    res = {}
    for prec in ("f16", "bf16"):
        c2f_grads(prec, res)
        awp_grads(prec, res)
    np.savez(sys.argv[1], **res)
    print("saved", len(res), "tensors to", sys.argv[1])
