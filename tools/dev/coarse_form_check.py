"""Developer check (GPU box): the 64-wide level's inference pass -- k_voxel_mlp_resident (default) against k_voxel_mlp_pipe (EVD_COARSE_FORM=pipe) -- bit for bit,
and its time.   python tools/dev/coarse_form_check.py <out.npy> [precision]    (run once per form: the switch is read once per process)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from evdeblurnerf_amd import _lib as L, weights as W
from evdeblurnerf_amd.voxnerf import VoxelNeRFRayFeatures

prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
nvox = 16777248
g = W.pdrf_grid_size(AABB[0], AABB[1], nvox)
sd = W.make_pdrf_state_dict(61, g, input_ch=95, hidden_dim=64, geo_feat_dim=15)
net = VoxelNeRFRayFeatures(sd, "", AABB, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, input_ch=95, app_dim=32, app_n_comp=(64, 16, 16), n_voxels=nvox,
                           precision=prec)
rs = np.random.RandomState(3)
for R, S in ((4096, 64), (333, 17)):
    pts = torch.tensor(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32), device="cuda")
    d = rs.normal(size=(R, 3)); vd = torch.tensor((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), device="cuda")
    fts = torch.tensor((0.3 * rs.normal(size=(R, S, 32))).astype(np.float32), device="cuda")
    z = torch.linspace(0, 1, S, device="cuda").expand(R, S).contiguous()
    rd = torch.tensor(d.astype(np.float32), device="cuda")
    # the C entry with feature = NULL, as the c2f render calls the coarse level (the Python wrapper always asks for the per-sample feature rows,
    # which only the generic kernel writes)
    f32 = dict(dtype=torch.float32, device="cuda")
    color, depth, acc, wts = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32), torch.empty((R, S), **f32)
    need = int(L.lib().evd_voxel_forward_workspace_bytes(net._h, R, S))
    ws = torch.empty((need,), dtype=torch.uint8, device="cuda")

    def fwd():
        L.check(L.lib().evd_voxel_forward(net._h, L.PREC[prec], L.ptr(pts), L.ptr(vd), 3, L.ptr(fts), 32, L.ptr(z), L.ptr(rd), 3, R, S, 0, L.ptr(color), L.ptr(depth),
                                          L.ptr(acc), L.ptr(wts), None, L.ptr(ws), need, L.stream_ptr()), "evd_voxel_forward")
        return color, depth, acc, wts
    out = fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = fwd()
    e1.record(); e1.synchronize()
    print(f"[{os.environ.get('EVD_COARSE_FORM', 'resident')}, {prec}] {R} x {S}: level forward (networks + compositing) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    if R == 4096:
        keep = [t.clone() for t in out]
arrs = [t.detach().cpu().numpy() for t in (keep if isinstance(keep, (tuple, list)) else [keep]) if torch.is_tensor(t)]
np.save(sys.argv[1], np.concatenate([a.reshape(-1).astype(np.float32) for a in arrs]))
