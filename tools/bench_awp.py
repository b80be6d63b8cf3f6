#!/usr/bin/env python
"""Timing of the AWP consumer's per-sample part at the blurfactory shape: fine level training forward + the sample
embedding (awp.py:98-100) + feature_integration (awp.py:102), forward and backward, two ways:
  torch : the level writes depth_feature [R P, S, 128] float32 (renderer.py:253-256), four torch Linear + ReLU, the scan kernel
  fused : the embedding reads the level's geo fragments (evd_awp_embed_forward / _backward), the scan kernel
GPU box only.   python tools/bench_awp.py [--rays 10240] [--samples 128] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.awp import SampleFeatureEmbed, feature_integration  # noqa: E402
from evdeblurnerf_amd.voxnerf import GeoFragments, VoxelNeRFSampleFeatures  # noqa: E402

AABB = ((-1.5, -1.5, -1.0), (1.5, 1.5, 1.0))


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=10240)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--prec", default="f16")
    a = ap.parse_args()
    R, S, dev = a.rays, a.samples, "cuda"
    nvox = 48 ** 3
    sd = W.make_pdrf_state_dict(71, W.pdrf_grid_size(AABB[0], AABB[1], nvox), input_ch=64 + 63, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
    net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=64 + 63, app_dim=32,
                                  app_n_comp=(64, 16, 16), n_voxels=nvox, precision=a.prec)
    flat = net.flat_params(sd)
    esd = W.make_awp_embed_state_dict(211)
    ws = [esd[f"sample_feature_embed_layer.{l}.weight"] for l in range(4)]
    bs = [esd[f"sample_feature_embed_layer.{l}.bias"] for l in range(4)]
    emb = SampleFeatureEmbed(ws, bs, precision=a.prec)
    eflat = torch.cat([torch.tensor(t).reshape(-1) for l in range(4) for t in (ws[l], bs[l])]).to(dev).requires_grad_(True)
    lin = torch.nn.ModuleList([torch.nn.Linear(128, 64)] + [torch.nn.Linear(64, 64) for _ in range(3)]).to(dev)
    rs = np.random.RandomState(0)
    pts = torch.tensor(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32), device=dev)
    d = rs.normal(size=(R, 3))
    vd = torch.tensor((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), device=dev)
    fts = torch.tensor((0.3 * rs.normal(size=(R, S, 64))).astype(np.float32), device=dev)
    z = torch.sort(torch.rand((R, S), device=dev), -1)[0]
    rd = torch.randn((R, 3), device=dev)
    gh = torch.randn((R, 64), device=dev) * 1e-3
    graw = torch.randn((R, S, 4), device=dev) * 1e-3

    def level_only():
        raw = net.mlp_train(flat, pts, vd, fts)
        (raw * graw).sum().backward()

    def torch_path():
        raw, feat = net.mlp_train(flat, pts, vd, fts, want_feature=True)
        h = feat
        for layer in lin:
            h = torch.relu(layer(h))
        hi = feature_integration(h.reshape(R, 1, S, 64), z, rd)
        ((raw * graw).sum() + (hi.reshape(R, 64) * gh).sum()).backward()

    def fused_path():
        geo = GeoFragments()
        raw, geo.token = net.mlp_train(flat, pts, vd, fts, want_feature=geo)
        h = emb(eflat, geo)
        hi = feature_integration(h.reshape(R, 1, S, 64), z, rd)
        ((raw * graw).sum() + (hi.reshape(R, 64) * gh).sum()).backward()

    t0, t1, t2 = timed(level_only, a.iters), timed(torch_path, a.iters), timed(fused_path, a.iters)
    n = R * S
    print(f"{a.prec} R={R} S={S} ({n / 1e6:.2f} M samples): fine level fwd+bwd alone {t0:.2f} ms | + AWP embed+scan, torch Linear on depth_feature "
          f"{t1:.2f} ms (+{t1 - t0:.2f}) | fused on the geo fragments {t2:.2f} ms (+{t2 - t0:.2f}); depth_feature tensor avoided: {n * 128 * 4 / 2**20:.0f} MiB "
          f"each way; embed store {emb and int(__import__('evdeblurnerf_amd')._lib.lib().evd_awp_embed_store_bytes(emb._h, n)) / 2**20:.0f} MiB")


if __name__ == "__main__":
    main()
