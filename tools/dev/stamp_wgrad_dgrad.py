"""Developer probe: where does a tile of k_wgrad_dgrad (the 256-wide level's fused wgrad + dgrad launches) spend its cycles?
Needs the -DEVD_WD_STAMP build (tools/dev/stamp_wgrad_dgrad.sh).  Runs one fine-level forward + backward at 10 240 rays x 128 samples.
    EVD_LIB_PATH=evdeblurnerf_amd/lib/variants/libevd_wdstamp.so python tools/dev/stamp_wgrad_dgrad.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from evdeblurnerf_amd import _lib as L, weights as W
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
nvox = 48 ** 3
g = W.pdrf_grid_size(AABB[0], AABB[1], nvox)
sd = W.make_pdrf_state_dict(32, g, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32, app_n_comp=(64, 16, 16),
                              n_voxels=nvox, precision="f16")
R, S = 10240, 128
rs = np.random.RandomState(0)
pts = torch.tensor(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32), device="cuda", requires_grad=True)
vd = torch.tensor(rs.normal(size=(R, 3)).astype(np.float32), device="cuda")
vd = (vd / vd.norm(dim=-1, keepdim=True)).requires_grad_(True)
fts = torch.tensor((0.3 * rs.normal(size=(R, S, 64))).astype(np.float32), device="cuda", requires_grad=True)
flat = net.flat_params(sd)
for _ in range(2):
    raw = net.mlp_train(flat, pts, vd, fts)
    (raw * torch.randn_like(raw) * 1e-3).sum().backward()
torch.cuda.synchronize()
lib = L.lib()
names = ["wait for the tile's DMA", "gradient forming (YGEN) + transposes + x exchange write", "barrier 1", "products (+ bias)", "dgrad + stores", "barrier 2"]
kinds = {8 + 8: "color_net.1 (YGEN, 8 x 8)", 8 + 16: "sigma_net.1 (4 + 1 row tiles x 8)", 5: "color_net.0 (8 x 5)", 4: "sigma_net.0 (8 x 4, rows out)"}
buf = (C.c_float * (2048 * 8))()
for kind, label in kinds.items():
    rc = lib.evd_debug_wd_stamps(buf, kind)
    assert rc == 0, rc
    t = np.ctypeslib.as_array(buf).reshape(2048, 8).astype(np.float64)
    t = t[t[:, 7] == -7]
    if not len(t):
        print(f"{label}: no stamps (library built without -DEVD_WD_STAMP, or the launch did not run)")
        continue
    tiles = t[:, 6].mean()
    tot = t[:, :6].sum(1).mean() / tiles
    print(f"{label}: {len(t)} wavefronts x {tiles:.1f} tiles; {tot:.0f} cycles per tile")
    for i, nme in enumerate(names):
        c = t[:, i].mean() / tiles
        print(f"    {nme:58s} {c:8.0f} cycles  {100 * c / tot:5.1f} %")
