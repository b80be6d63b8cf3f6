"""Developer probe: where does a pass of k_voxel_mlp_c (PDRF fine level, compensated float16) spend its cycles?  Needs a library whose
kernel_voxel_pipe_f16c.hip / kernel_voxel_train_f16c.hip were built with -DEVD_C_STAMP (tools/dev/stamp_voxel_c.sh): lanes 0..3 of
every wavefront write shader-clock stamps in place of their samples.   EVD_LIB_PATH=... python tools/dev/stamp_voxel_c.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from evdeblurnerf_amd import _lib as L, weights as W
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
nvox = 48 ** 3
gsz = W.pdrf_grid_size(AABB[0], AABB[1], nvox)
sd = W.make_pdrf_state_dict(71, gsz, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32,
                              app_n_comp=(64, 16, 16), n_voxels=nvox, precision="f16c")
R, S = 4096, 128
rs = np.random.RandomState(0)
T = lambda a: torch.as_tensor(a, device="cuda")
pts = T(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32))
d = rs.normal(size=(R, 3)); vd = T((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32))
fts = T((0.3 * rs.normal(size=(R, S, 64))).astype(np.float32))
z = torch.sort(torch.rand((R, S), device="cuda"), -1)[0]
n = R * S
names = ["inputs+encode", "start_wait+prime", "L0", "Sigma", "Geo", "C0", "C1", "C2", "output"]


def report(tag, raw):
    r = raw.reshape(-1, 32, 4)[:, :4].cpu().numpy().astype(np.float64)      # lanes 0..3 of every wavefront
    assert (r[:, 2, 3] == -7).all(), "library was not built with -DEVD_C_STAMP"
    seg = np.concatenate([r[:, 0], r[:, 1], r[:, 2, :1]], -1)               # [waves, 9]
    tot = seg.sum(-1)
    print(f"[{tag}] {len(r)} wavefront passes; cycles per pass: mean {tot.mean():.0f}, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:18s} {seg[:, i].mean():8.0f} cycles  {100 * seg[:, i].mean() / tot.mean():5.1f} %")
    print(f"   (inside the layers: vmcnt waits {r[:, 2, 1].mean():.0f}, barriers {r[:, 2, 2].mean():.0f} cycles per pass)")
    t0 = r[:, 3, 0] + r[:, 3, 1] * (1 << 24)
    blk, tile = r[:, 3, 2], r[:, 3, 3]
    # gap between consecutive passes of one workgroup (wave 0 of the block): start(next) - start(this) - this pass's length
    w0 = np.arange(len(r)) % 4 == 0
    order = np.lexsort((tile[w0], blk[w0]))
    b, tt, st, ln = blk[w0][order], tile[w0][order], t0[w0][order], tot[w0][order]
    same = b[1:] == b[:-1]
    if same.any():
        per = (st[1:] - st[:-1])[same]
        print(f"   start-to-start of consecutive passes of a workgroup: mean {per.mean():.0f} cycles (pass length {ln.mean():.0f})")


raw_t, store, _ = net.mlpforward_train(pts, vd, fts, "f16c")
for _ in range(3):
    raw_t, store, _ = net.mlpforward_train(pts, vd, fts, "f16c")
torch.cuda.synchronize()
report("TRAIN", raw_t)
need = int(L.lib().evd_voxel_forward_workspace_bytes(net._h, R, S))
ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
color, depth, acc, wts = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32), torch.empty((R, S), **f32)
rd = torch.randn((R, 3), device="cuda")
for _ in range(4):
    L.check(L.lib().evd_voxel_forward(net._h, L.PREC["f16c"], L.ptr(pts), L.ptr(vd), 3, L.ptr(fts), 64, L.ptr(z), L.ptr(rd), 3, R, S, 0, L.ptr(color), L.ptr(depth),
                                      L.ptr(acc), L.ptr(wts), None, L.ptr(ws), need, L.stream_ptr()), "evd_voxel_forward")
torch.cuda.synchronize()
off = (-ws.data_ptr()) % 256
report("inference", ws[off:off + n * 16].view(torch.float32))
