"""developer probe: per-sample cost of the fine level's MLP backward (dgrad + wgrad chain) against the batch size -- does a batch whose
gradient fragments fit the 256 MB Infinity Cache run the chain faster per sample (fragments written by one kernel, read by the next two)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures
AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
nvox = 48 ** 3
sd = W.make_pdrf_state_dict(71, W.pdrf_grid_size(AABB[0], AABB[1], nvox), input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32, app_n_comp=(64, 16, 16), n_voxels=nvox, precision="f16")
flat = net.flat_params(sd)
net.load_params(flat)
S = 128
for R in (512, 1024, 2048, 4096, 8192, 18432):
    rs = np.random.RandomState(0)
    pts = torch.as_tensor(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32), device="cuda")
    d = rs.normal(size=(R, 3)); vd = torch.as_tensor((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), device="cuda")
    fts = torch.as_tensor((0.3 * rs.normal(size=(R, S, 64))).astype(np.float32), device="cuda")
    d_raw = torch.randn((R, S, 4), device="cuda") * 1e-3
    raw, store, _ = net.mlpforward_train(pts, vd, fts, "f16")
    acc = torch.zeros_like(flat)
    def bwd():
        net.mlp_backward_flat(d_raw, raw, store, "f16", want_fts=True, pts=pts, viewdirs=vd, accumulate_into=acc)
    for _ in range(3): bwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): bwd()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"R={R:6d}: {R * S / 1e6:5.2f} M samples, store {store.numel() / 2**20:7.0f} MiB, backward {ms:7.3f} ms = {ms / (R * S) * 2**19:.3f} ms per 2^19 samples")
    del store, raw, pts, fts, d_raw
